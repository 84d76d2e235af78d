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

from . import models, opts, synthetic
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


def checkpoint(opt, model, optimizer_dict, iteration, epoch, loss_history, loader=None):
    """train.py:299-336 (save_any_kind_of_model, save_optimizer, dump_infos_miscellaneous, save_results): weights,
    optimizer states, and the infos / histories records as JSON (the reference pickles them; nothing here is ever
    unpickled).  Besides the reference's fields the infos record carries what this implementation needs to continue
    the SAME run: the schedule values in force and the position of the noise stream and of the loader."""
    if dist.is_initialized() and dist.get_rank() != 0:
        return
    save_model(model, opt, 'alternatingModel' if opt.is_alternating else 'model', iteration)
    save_optimizer(opt, optimizer_dict)
    cg = model.caption_generator
    infos = dict(iter=iteration, epoch=epoch, gumbel_temp=float(cg.gumbel_temp),
                 ss_prob=float(cg.ss_prob), current_lr=float(getattr(opt, 'current_lr', opt.learning_rate)),
                 retrieval_reward_weight=float(model.retrieval_reward_weight),
                 prob_gumbel_softmax=float(cg.prob_gumbel_softmax), prob_multinomial_soft=float(cg.prob_multinomial_soft),
                 noise=dict(seed=int(cg.noise.seed), counter=int(cg.noise.counter)),
                 iterators=_loader_state(loader),
                 opt={k: v for k, v in vars(opt).items() if isinstance(v, (int, float, str, list, type(None)))})
    for name in ('infos_' + opt.id + '.json', f'infos_{opt.id}-{iteration}.json'):       # save_pkl writes both (:109-118)
        with open(os.path.join(opt.checkpoint_path, name), 'w') as f:                     # JSON, not pickle
            json.dump(infos, f)
    with open(os.path.join(opt.checkpoint_path, 'histories_' + opt.id + '.json'), 'w') as f:
        json.dump(dict(loss_history=loss_history), f)


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
    if infos.get('current_lr') is not None:
        opt.current_lr = infos['current_lr']
        for o in _all_optimizers(optimizer_dict):
            utils.set_lr(o, opt.current_lr)
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        from .optimizer import overlap_gradient_exchange
        overlap_gradient_exchange(model, optimizer_dict)
    update_lr_flag = True
    epoch, epoch_start, iteration = load_from_infos(infos, None, opt)
    reported_path = False
    loss_history = {}
    num_turns = len(opt.alternating_turn) if opt.is_alternating else 1
    init_scorer(opt.cached_tokens)
    while True:
        curr_turn = opt.alternating_turn[iteration % num_turns] if opt.is_alternating else 'optimizer'
        optimizer = optimizer_dict[curr_turn]
        update_lr_flag = apply_schedules(update_lr_flag, opt, epoch, optimizer_dict, optimizer, model, epoch_start,
                                         iteration)
        start = time.time()
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
        train_loss = float(loss.detach())                       # the step's one host sync (train.py:533-535)
        end = time.time()
        if rank == 0 and not reported_path and hasattr(model, 'caption_generator'):
            fused = getattr(model.caption_generator, 'last_pair_fused', None)
            if fused is not None:      # which decode path this batch size gets (B % 32 == 0 and B <= 128: shared launches)
                print('sampled + greedy decodes: ' + ('one launch chain over 2B rows' if fused else 'two launch chains'))
                reported_path = True
        if rank == 0:
            extra = ' '.join(f'{k} = {float(v):.3f}' for k, v in model.loss().items())
            print(f'iter {iteration} (epoch {epoch}) [{curr_turn}], train_loss = {train_loss:.4f}, '
                  f'time/batch = {end - start:.4f}  {extra}', flush=True)
        iteration += 1
        if data['bounds']['wrapped']:
            epoch += 1
            update_lr_flag = True
        if iteration % opt.losses_log_every == 0:
            loss_history[iteration] = train_loss
        if iteration % opt.save_checkpoint_every == 0:
            checkpoint(opt, model, optimizer_dict, iteration, epoch, loss_history, loader)
        if (epoch >= opt.max_epochs != -1) or (0 < opt.max_iterations <= iteration):
            break
    if hasattr(loader, 'close'):
        loader.close()
    return model


def main(argv=None):
    train(opts.parse_opt(argv))


if __name__ == '__main__':
    main()
