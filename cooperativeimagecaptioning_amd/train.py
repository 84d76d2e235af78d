"""Training driver with the reference's loop structure (train.py:473-562): per iteration pick the
turn, apply the epoch schedules, fetch a batch, zero, forward+backward, clamp+Adam, log, and
checkpoint every --save_checkpoint_every iterations.

The dataset reader of the reference (dataloader.py: h5py/lmdb + 4 worker processes) is host I/O
outside the hot path; this driver consumes any object with the same ``get_batch('train')``
output contract (dataloader.py:171-245).  ``--synthetic 1`` selects COCO-shaped synthetic
batches (synthetic.SyntheticLoader), which is what the benchmark and smoke runs use.
"""
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import models, opts, status, synthetic
from .misc import utils
from .misc.rewards import init_scorer
from .optimizer import load_optimizer, save_optimizer, zeroing_optimizer, update_optimizer


# ---- schedules (train.py:26-92,399-435) -------------------------------------------------------
def polynomial_decay(epoch, decay_factor, power, initial_rate=1):
    return np.minimum(initial_rate, initial_rate * 1 / (decay_factor * (1 + epoch)) ** power)


def update_learning_rate(opt, epoch, optimizer_dict, optimizer):
    if epoch > opt.learning_rate_decay_start >= 0:
        frac = (epoch - opt.learning_rate_decay_start) // opt.learning_rate_decay_every
        opt.current_lr = opt.learning_rate * opt.learning_rate_decay_rate ** frac
        for o in _all_optimizers(optimizer_dict):
            utils.set_lr(o, opt.current_lr)
    else:
        opt.current_lr = opt.learning_rate


def _all_optimizers(optimizer_dict):
    for v in optimizer_dict.values():
        if isinstance(v, dict):
            yield from v.values()
        else:
            yield v


def apply_schedules(update_lr_flag, opt, epoch, optimizer_dict, optimizer, model, epoch_start, iteration):
    if update_lr_flag:
        update_learning_rate(opt, epoch, optimizer_dict, optimizer)
        if epoch > opt.scheduled_sampling_start >= 0:                              # train.py:80-85
            frac = (epoch - opt.scheduled_sampling_start) // opt.scheduled_sampling_increase_every
            opt.ss_prob = min(opt.scheduled_sampling_increase_prob * frac, opt.scheduled_sampling_max_prob)
            model.caption_generator.ss_prob = opt.ss_prob
        if epoch > opt.retrieval_reward_weight_decay_start >= 0:                   # train.py:88-92
            frac = (epoch - opt.retrieval_reward_weight_decay_start) // opt.retrieval_reward_weight_decay_every
            model.retrieval_reward_weight = opt.retrieval_reward_weight * opt.retrieval_reward_weight_decay_rate ** frac
        update_lr_flag = False
    if opt.softmax_cooling_decay_factor > 0:                                       # train.py:32-47
        prob = 1 - polynomial_decay(epoch - epoch_start, opt.softmax_cooling_decay_factor, power=0.5)
        if opt.retrieval_reward == 'multinomial_soft':
            model.caption_generator.prob_multinomial_soft = prob
        elif opt.retrieval_reward == 'gumbel_softmax':
            model.caption_generator.prob_gumbel_softmax = prob
    if opt.gumbel_temperature_annealing_factor > 0 and iteration % opt.num_iteration_for_annealing == 0:
        frac = max(0.5, np.exp(-opt.gumbel_temperature_annealing_factor * (iteration - 177000)))   # train.py:399-414
        model.caption_generator.gumbel_temp = model.caption_generator.gumbel_temp * frac
    return update_lr_flag


# ---- checkpoints (train.py:95-129,299-347) ----------------------------------------------------
def save_model(model, opt, model_kind, iteration=None):
    os.makedirs(opt.checkpoint_path, exist_ok=True)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    torch.save(sd, os.path.join(opt.checkpoint_path, model_kind + '.pth'))
    if iteration:
        torch.save(sd, os.path.join(opt.checkpoint_path, f'{model_kind}-{iteration}.pth'))


def _jsonable(x):
    """Histories / infos records as plain JSON: numpy scalars and arrays, tensors and integer keys made portable."""
    if isinstance(x, dict):
        return {str(k): _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if torch.is_tensor(x):
        return x.item() if x.numel() == 1 else x.tolist()
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    return x


def save_json(args, file_name, save_me, iteration=None, best=None):
    """train.py:93-116 (save_pkl) with the reference's three file names - <name>_<id>.json, <name>_<id>-<iteration>.json,
    <name>_<id>-best.json - written as JSON: nothing this trainer writes ever needs unpickling."""
    assert not (iteration is not None and best is not None), 'Only one of (iteration, best) can be different than None'
    ext = f'-{iteration}' if iteration else ('-best' if best else '')
    path = os.path.join(args['checkpoint_path'], f"{file_name}_{args['id']}{ext}.json")
    with open(path, 'w') as f:
        json.dump(_jsonable(save_me), f)
    print(f"{file_name}_{args['id']}{ext}.json saved to {args['checkpoint_path']}")


def load_histories(opt):
    """train.py:187-195: the histories record of the run to continue (JSON; integer keys restored)."""
    histories = {}
    if vars(opt).get('start_from', None) is not None:
        path = os.path.join(opt.start_from, 'histories_' + opt.id + '.json')
        if os.path.isfile(path):
            with open(path) as f:
                raw = json.load(f)
            histories = {k: ({int(i): v for i, v in d.items()} if isinstance(d, dict) else d) for k, d in raw.items()}
    return histories


def evaluate_model(opt, model, loader, iteration, val_result_history):
    """train.py:238-251: validation losses, one generated caption per image and (with --rank_eval 1) the retrieval ranks of
    the val split, on the device engines (eval_utils.eval_split -> greedy / beam decode, cic_listener_fwd,
    cic_retrieval_ranks)."""
    from . import eval_utils
    eval_kwargs = {'split': 'val', 'dataset': opt.input_json, 'use_att': opt.use_att}
    eval_kwargs.update(vars(opt))
    eval_kwargs.setdefault('verbose', False)
    val_loss, predictions, lang_stats = eval_utils.eval_split(model, loader, eval_kwargs,
                                                              useGenSent=opt.rank_on_gen_captions)
    val_result_history[iteration] = {'loss': val_loss, 'lang_stats': lang_stats, 'predictions': predictions}
    return val_result_history, lang_stats, val_loss


def get_current_score(opt, lang_stats, val_loss):
    """train.py:254-277.  language_eval (CIDEr from the Java coco-caption pipeline) is out of scope: the speaker's
    selection score is -loss_cap, as the reference computes it with --language_eval 0; the listener's is
    100 x val_loss[vse_eval_criterion] (rsum of the rank evaluation with --rank_eval 1, else 0)."""
    if opt.language_eval == 1:
        raise NotImplementedError('language_eval = 1 selects the best model by CIDEr from the Java coco-caption pipeline '
                                  '(train.py:256-263): out of scope, run with --language_eval 0')
    current_score = 0 if opt.phase == 1 else -val_loss.get('loss_cap', 0.0)
    crit = val_loss.get(opt.vse_eval_criterion, None)
    if crit is None and isinstance(val_loss.get('val'), dict):               # phase 1 ranks come per split (:274-276)
        crit = val_loss['val'].get(opt.vse_eval_criterion, 0)
    return float(current_score), float(crit or 0) * 100


def check_if_best(current_score, best_val_score, current_score_vse, best_val_score_vse):
    """train.py:280-292."""
    best_flag = best_flag_vse = False
    if best_val_score is None or current_score > best_val_score:
        best_val_score, best_flag = current_score, True
    if best_val_score_vse is None or current_score_vse > best_val_score_vse:
        best_val_score_vse, best_flag_vse = current_score_vse, True
    return best_val_score, best_flag, best_val_score_vse, best_flag_vse


def operations_in_checkpoint(opt, model, loader, iteration, epoch, best, optimizer_dict, infos, histories, eval_loader=None):
    """train.py:438-470: evaluate on the val split, score, save weights + optimizers + infos + histories, and keep
    `model-best.pth` / `model_vse-best.pth` with their infos copies when a selection score improved.

    best: {'score', 'score_vse'} - the best validation scores so far, UPDATED in place.  (The reference passes the two
    scores by value and never hands the new ones back to its loop, train.py:543-547, so inside one run it keeps comparing
    with the scores it started with and only a restart picks the recorded ones up from infos; the running best is what the
    record and the -best files are for, so it is kept here.)
    Records are JSON (the reference pickles them).  Besides the reference's fields the infos record carries what this
    implementation needs to continue the SAME run: the schedule values in force, the noise stream's and the loader's
    position."""
    if dist.is_initialized() and dist.get_rank() != 0:
        return
    val_result_history = histories.setdefault('val_result_history', {})
    ev = eval_loader if eval_loader is not None else getattr(loader, 'loader', loader)
    flags = (False, False)
    if ev is not None and getattr(opt, 'eval_at_checkpoint', 1) and hasattr(ev, 'reset_iterator'):
        # the evaluation draws from the speaker's noise stream (sampled captions of the loss forward): the TRAINING stream
        # continues where it was, so a run's weights do not depend on how often it was evaluated (and a resume is exact)
        noise_at = model.caption_generator.noise.counter
        val_result_history, lang_stats, val_loss = evaluate_model(opt, model, ev, iteration, val_result_history)
        model.caption_generator.noise.counter = noise_at
        score, score_vse = get_current_score(opt, lang_stats, val_loss)
        best['score'], f1, best['score_vse'], f2 = check_if_best(score, best.get('score'), score_vse, best.get('score_vse'))
        flags = (f1, f2)
        print(f'validation at iteration {iteration}: score {score:.4f} (best {best["score"]:.4f}), '
              f'listener score {score_vse:.2f} (best {best["score_vse"]:.2f})')
    save_model(model, opt, 'alternatingModel' if opt.is_alternating else 'model', iteration)      # :295-302
    save_optimizer(opt, optimizer_dict)
    cg = model.caption_generator
    infos.update(iter=iteration, epoch=epoch, iterators=_loader_state(loader),                    # :305-317
                 best_val_score=best.get('score'), best_val_score_vse=best.get('score_vse'),
                 vocab=(ev.get_vocab() if hasattr(ev, 'get_vocab') else None), gumbel_temp=float(cg.gumbel_temp),
                 opt={k: v for k, v in vars(opt).items() if isinstance(v, (int, float, str, list, type(None)))},
                 ss_prob=float(cg.ss_prob), current_lr=float(getattr(opt, 'current_lr', opt.learning_rate)),
                 retrieval_reward_weight=float(model.retrieval_reward_weight),
                 prob_gumbel_softmax=float(cg.prob_gumbel_softmax), prob_multinomial_soft=float(cg.prob_multinomial_soft),
                 noise=dict(seed=int(cg.noise.seed), counter=int(cg.noise.counter)))
    args = {'checkpoint_path': opt.checkpoint_path, 'id': opt.id}
    save_json(args, 'infos', infos)                                                               # save_results :333-336
    save_json(args, 'infos', infos, iteration=iteration)
    save_json(args, 'histories', histories)
    if flags[0]:                                                                                  # save_best_results :339-347
        save_model(model, opt, 'model-best')
        save_json(args, 'infos', infos, best=True)
    if flags[1]:
        save_model(model, opt, 'model_vse-best')
        save_json(args, 'infos_vse', infos, best=True)


def _loader_state(loader):
    """Position of the batch source, as the reference keeps loader.iterators / split_ix in infos (train.py:312-313)."""
    if loader is None:
        return None
    # PrefetchLoader: the batches it pulled ahead of the step are replayed after a resume (its state_dict() hands back the
    # snapshot of the oldest one)
    return loader.state_dict() if hasattr(loader, 'state_dict') else None


def load_infos(opt):
    """train.py:143-159: the infos record of the run to continue ({} for a fresh run), after checking that the saved
    model options agree with the command line.  Only this repository's own JSON record is read."""
    infos = {}
    if vars(opt).get('start_from', None) is not None:
        filename = os.path.join(opt.start_from, 'infos_' + opt.id + '.json')
        if not os.path.isfile(filename):
            if os.path.isfile(filename[:-5] + '.pkl'):
                print(f'{filename[:-5]}.pkl is a pickle of the reference implementation: not loaded (pickles execute code); '
                      f'schedules restart from iteration 0')
            return infos
        with open(filename) as f:
            print('read from [%s]' % filename)
            infos = json.load(f)
        saved = infos.get('opt', {})
        for checkme in ('caption_model', 'rnn_type', 'rnn_size', 'num_layers'):
            assert saved.get(checkme) == vars(opt).get(checkme), \
                "Command line argument and saved model disagree on '%s' " % checkme
    return infos


def load_from_infos(infos, loader, opt):
    """train.py:360-367."""
    iteration = infos.get('iter', 0)
    epoch = infos.get('epoch', 0)
    epoch_start = epoch
    st = infos.get('iterators')
    inner = getattr(loader, 'loader', loader)
    if loader is not None and st is not None and hasattr(inner, 'load_state_dict'):
        inner.load_state_dict(st)
    return epoch, epoch_start, iteration


class LossLog:
    """The per-iteration log line without draining the queue.  The reference reads `loss.data[0]` right after the step
    (train.py:533-535): a host sync per iteration, after which the device idles while Python prepares the next step's first
    launches.  Here the step's loss and the logged terms (model.loss(): 0-dim device values) are gathered by ONE small launch,
    copied to page-locked memory asynchronously and read when their event has fired - normally while the NEXT step is being
    enqueued; the host never runs more than `depth` iterations ahead of the device.

    The device's sticky status word (status.py: a hand-off of a one-launch recurrence timed out) rides in the same copy, its
    32 bits carried unchanged in a float slot: pop() raises CicError naming the loop and the iteration whose line it was
    reading - the first line after the failure - before that line is emitted.  The optimiser kernels read the same word on
    the device and skip their updates from the failing step on, so the weights stay those of the last good step."""

    def __init__(self, device, depth=2, width=32):
        self.depth, self.width = depth, width
        self.slots = [torch.empty(width, dtype=torch.float32).pin_memory() for _ in range(depth + 1)]
        self.pending = []                   # [(meta, keys, slot, event)]
        self.last = {}                      # iteration -> train_loss of the lines already read
        self._n = 0

    def push(self, meta, loss, terms):
        keys = list(terms.keys())[:self.width - 2]
        vals = [loss.detach().reshape(())] + [torch.as_tensor(terms[k], device=loss.device).detach().reshape(()).float()
                                              for k in keys]
        vals.append(status.word(loss.device).view(torch.float32)[0])      # bit pattern of the int32 word (copied, never computed on)
        slot = self.slots[self._n % len(self.slots)]
        self._n += 1
        slot[:len(vals)].copy_(torch.stack(vals), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((meta, keys, slot, ev, len(vals)))

    def pop(self, emit, block_to=None):
        """Emit every line whose values have landed; block until at most `block_to` lines are outstanding."""
        block_to = self.depth if block_to is None else block_to
        while self.pending and (len(self.pending) > block_to or self.pending[0][3].query()):
            meta, keys, slot, ev, n = self.pending.pop(0)
            ev.synchronize()
            status.raise_if_set(int(slot[n - 1:n].view(torch.int32)[0]), f"seen with the log line of iteration {meta['iteration']}")
            vals = slot[:n - 1].tolist()
            self.last[meta['iteration']] = vals[0]
            emit(meta, vals[0], dict(zip(keys, vals[1:])))


def load_data(data, opt, device):
    """train.py:162-178: host batch -> device tensors."""
    tens = lambda x: None if x is None else torch.as_tensor(x).to(device, non_blocking=True)   # noqa: E731
    return tens(data['fc_feats']), tens(data.get('att_feats')), tens(data.get('att_masks')), \
        tens(data['labels']), tens(data['masks'])


def train(opt, loader=None):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL over xGMI, one rank per GPU.  CIC_DIST_BACKEND=gloo CIC_SAME_DEVICE=1 rehearses the multi-process loop on a
        # box with fewer GPUs than ranks (every rank on cuda:0, exchanges through the host)
        backend = os.environ.get('CIC_DIST_BACKEND', 'nccl')
        if os.environ.get('CIC_SAME_DEVICE') == '1':
            assert backend == 'gloo', 'CIC_SAME_DEVICE is a gloo rehearsal (RCCL wants one GPU per rank)'
            from . import engine as _engine
            _engine.DEVICE_SHARED[0] = True           # several ranks compute on one GPU
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend)
    device = torch.device('cuda', local_rank)
    opt.use_att = utils.if_use_att(opt)
    made_synthetic = False
    if loader is None:
        if opt.synthetic:
            made_synthetic = True
            opt.vocab_size = getattr(opt, 'vocab_size', None) or 9487
            opt.seq_length = getattr(opt, 'seq_length', None) or 16
            loader = synthetic.SyntheticLoader(opt, seed=1234 + rank, pool=getattr(opt, 'synthetic_pool', 8))
        else:
            from .dataloader import DataLoader       # the reference's on-disk formats, read with numpy (dataloader.py)
            loader = DataLoader(opt)
    if not made_synthetic:
        opt.vocab_size, opt.seq_length = loader.vocab_size, loader.seq_length
    infos = load_infos(opt)                                   # {} for a fresh run
    if infos.get('iterators') is not None and hasattr(loader, 'load_state_dict'):
        loader.load_state_dict(infos['iterators'])            # before the prefetcher pulls its first batch
    if getattr(opt, 'prefetch', 1):
        from .prefetch import PrefetchLoader          # the next batch is uploaded on a copy stream under this step's compute
        loader = PrefetchLoader(loader, device, 'train')
    opt.gumbel_temp = infos.get('gumbel_temp', opt.gumbel_temp)                # train.py:366, before the model reads it
    torch.manual_seed(opt.seed)                               # same initial weights on every rank
    model = models.AlternatingJointModel(opt).to(device).train()   # loads <start_from>/*.pth when asked to continue
    if getattr(opt, 'share_embed', 0):                            # train.py:390-391: re-tie after the move to the device
        model.tie_embeddings()
    cg = model.caption_generator
    cg.noise.manual_seed(opt.seed * 1000 + rank)
    if infos:                                                 # the schedule values and stream positions in force at the checkpoint
        cg.ss_prob = infos.get('ss_prob', cg.ss_prob)
        cg.prob_gumbel_softmax = infos.get('prob_gumbel_softmax', cg.prob_gumbel_softmax)
        cg.prob_multinomial_soft = infos.get('prob_multinomial_soft', cg.prob_multinomial_soft)
        model.retrieval_reward_weight = infos.get('retrieval_reward_weight', model.retrieval_reward_weight)
        if infos.get('noise'):
            cg.noise.counter = int(infos['noise']['counter'])   # every rank keeps its own seed and continues its stream
    optimizer_dict = load_optimizer(model, opt)
    from .optimizer import fuse_zero_grad
    fuse_zero_grad(optimizer_dict)                            # gradients are cleared inside the clamp+Adam kernels
    if infos.get('current_lr') is not None:
        opt.current_lr = infos['current_lr']
        for o in _all_optimizers(optimizer_dict):
            utils.set_lr(o, opt.current_lr)
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        from .optimizer import overlap_gradient_exchange
        overlap_gradient_exchange(model, optimizer_dict)
    update_lr_flag = True
    epoch, epoch_start, iteration = load_from_infos(infos, None, opt)
    histories = load_histories(opt)                                                # train.py:374-379
    loss_history = histories.setdefault('loss_history', {})
    lr_history = histories.setdefault('lr_history', {})
    ss_prob_history = histories.setdefault('ss_prob_history', {})
    best = {'score': None, 'score_vse': None}
    if getattr(opt, 'load_best_score', 1):                                         # train.py:369-374
        best = {'score': infos.get('best_val_score'), 'score_vse': infos.get('best_val_score_vse')}
    reported_path = False
    num_turns = len(opt.alternating_turn) if opt.is_alternating else 1
    init_scorer(opt.cached_tokens)
    log = LossLog(device)

    def emit(meta, train_loss, terms):
        if meta['to_history']:                                                     # write_loss_summary, train.py:229-235
            loss_history[meta['iteration'] + 1] = train_loss
        if rank == 0:
            extra = ' '.join(f'{k} = {v:.3f}' for k, v in terms.items())
            print(f"iter {meta['iteration']} (epoch {meta['epoch']}) [{meta['turn']}], train_loss = {train_loss:.4f}, "
                  f"time/batch = {meta['host_s']:.4f}  {extra}", flush=True)

    last_end = time.time()
    while True:
        curr_turn = opt.alternating_turn[iteration % num_turns] if opt.is_alternating else 'optimizer'
        optimizer = optimizer_dict[curr_turn]
        update_lr_flag = apply_schedules(update_lr_flag, opt, epoch, optimizer_dict, optimizer, model, epoch_start,
                                         iteration)
        data = loader.get_batch('train')
        fc_feats, att_feats, att_masks, labels, masks = load_data(data, opt, device)
        zeroing_optimizer(opt, optimizer_dict, optimizer)
        if opt.is_alternating:
            loss = model(fc_feats, labels, masks, data, att_feats, att_masks, is_alternating=True,
                         alternating_turn=curr_turn)
        else:
            loss = model(fc_feats, labels, masks, data, att_feats, att_masks)
        loss.backward()
        update_optimizer(optimizer_dict, optimizer, opt)
        if hasattr(loader, 'prefetch'):
            loader.prefetch()                                   # batch i+1 travels to HBM while step i computes
        end = time.time()
        # the loss and the logged terms follow asynchronously; the line of iteration i is printed once they have landed
        # (while step i+1 is being enqueued): steady-state time per iteration = what the device needs, as in bench.py
        start_ckpt = bool(getattr(opt, 'start_with_checkpoint', 0))
        to_history = (iteration + 1) % opt.losses_log_every == 0 or start_ckpt     # train.py:540-543
        log.push(dict(iteration=iteration, epoch=epoch, turn=curr_turn, host_s=end - last_end, to_history=to_history),
                 loss, model.loss())
        last_end = end
        log.pop(emit)
        if rank == 0 and not reported_path and hasattr(model, 'caption_generator'):
            fused = getattr(model.caption_generator, 'last_pair_fused', None)
            if fused is not None:      # which decode path this batch size gets (B % 32 == 0 and B <= 128: shared launches)
                print('sampled + greedy decodes: ' + ('one launch chain over 2B rows' if fused else 'two launch chains'))
                reported_path = True
        iteration += 1
        if data['bounds']['wrapped']:
            epoch += 1
            update_lr_flag = True
        if to_history:
            lr_history[iteration] = float(getattr(opt, 'current_lr', opt.learning_rate))
            ss_prob_history[iteration] = float(model.caption_generator.ss_prob)
        done = (epoch >= opt.max_epochs != -1) or (0 < opt.max_iterations <= iteration)
        if iteration % opt.save_checkpoint_every == 0 or start_ckpt:               # train.py:546-552
            log.pop(emit, block_to=0)                            # the histories record holds every line up to here
            status.check(device, f'before the checkpoint of iteration {iteration}')   # never save weights behind a failed hand-off
            operations_in_checkpoint(opt, model, loader, iteration, epoch, best, optimizer_dict, infos, histories)
            model.train()
        if start_ckpt:
            opt.start_with_checkpoint = 0
        if done:
            break
    log.pop(emit, block_to=0)
    status.check(device, 'at the end of training')
    if hasattr(loader, 'close'):
        loader.close()
    return model


def main(argv=None):
    train(opts.parse_opt(argv))


if __name__ == '__main__':
    main()
