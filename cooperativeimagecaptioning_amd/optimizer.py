"""Mirror of the reference's optimizer.py (define/load/zero/update/save, :25-242) on top of
FlatAdam: Adam with torch's default betas/eps (the reference ignores its --optim* flags,
optimizer.py:25-27) whose step is ONE fused clamp+Adam kernel over the agent's flat parameter
buffer (cic_clamp_adam), preceded — when torch.distributed is initialised — by ONE RCCL
all-reduce of the agent's flat gradient buffer (data-parallel replicas).
"""
import os

import torch
import torch.distributed as dist

from . import engine
from .misc import utils


class FlatAdam:
    """torch.optim.Adam look-alike for one agent (a module owning a FlatAgent via .flat()).

    Data-parallel exchange: the flat gradient buffer is cut into contiguous BUCKETS, each summed over the ranks by one
    all-reduce before the clamp.  A bucket whose gradient is final early can be started from inside backward()
    (begin_all_reduce) and travels under the rest of the backward pass; step() starts whatever has not been started,
    waits for all of them and folds 1/world into the Adam kernel.  The speaker has two buckets — 'logit' (the logit
    layer, 19.4 MB, final before the BPTT loop; laid out last in the flat buffer) and 'rest' —, the listener one."""

    def __init__(self, module, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        if weight_decay != 0:
            # torch.optim.Adam (the reference, optimizer.py:25-27) skips parameters whose grad is None and keeps a step
            # count per parameter; the flat update decays and bias-corrects every parameter alike.  The two agree at the
            # reference's default weight_decay = 0 (a zero gradient leaves a zero-moment parameter where it is).
            raise NotImplementedError('weight_decay != 0 is not on the MI355X path (FlatAdam updates the whole flat '
                                      'buffer; parameters without gradient would decay, unlike torch.optim.Adam)')
        self.module = module
        self.param_groups = [dict(params=list(module.parameters()), lr=lr, weight_decay=weight_decay,
                                  betas=betas, eps=eps)]
        self._grad_clip = None
        self.grad_scale = 1.0         # extra factor on the gradient (1/n after accumulating n micro-batches)
        self._pending = {}            # bucket name -> async all-reduce handle of THIS step
        self._done = set()            # buckets already summed over the ranks in this step
        # gradient accumulation under data parallelism: while True, the early starts from inside backward() are skipped
        # (the gradient of a bucket is final only after the LAST micro-batch's backward); step() exchanges whatever has
        # not left.  See accumulate_gradients().
        self.defer_exchange = False
        # True: step() clears the gradient buffer inside the clamp+Adam kernel (each element is in registers there) and the
        # zero_grad() that opens the next step finds nothing to do.  Gradients are then NOT readable after step() - the
        # trainer and the benchmark never do; off by default, as torch.optim keeps .grad until zero_grad().
        self.zero_grad_in_step = False
        self._grad_is_zero = False
        # share_embed = 1: parameters of this agent that live in another agent's flat buffers (flat.FlatAgent.external) - the one
        # embedding table both Adam instances own (AlternatingJointModel.py:83-88, optimizer.py:25-27).  This optimizer keeps
        # moments of its own for them and steps them with a launch of their own; the OWNER's optimizer steps them again with
        # its moments, from the same gradient (optimizer.py:233-242: the clamp is idempotent).  name -> (exp_avg, exp_avg_sq)
        self._ext_state = {}
        self.ext_owner = None         # the owner's FlatAdam (set by load_optimizer): its exchange must have landed before our step

    @property
    def flat(self):
        return self.module.flat()

    def buckets(self):
        """{name: (start, end)} over the flat gradient buffer, covering it exactly once."""
        fl = self.flat
        if fl.tail_offset < fl.numel:
            return {'rest': (0, fl.tail_offset), 'logit': (fl.tail_offset, fl.numel)}
        return {'all': (0, fl.numel)}

    def _drain(self):
        for h in self._pending.values():
            h.wait()
        self._pending.clear()

    def _externals(self):
        fl = self.flat
        return [(n, p) for n, p, o in zip(fl.names, fl.params, fl.offsets) if o is None]

    def zero_grad(self, set_to_none=False):
        # an exchange still in flight belongs to a step that never reached step() (a skipped or failed update, an
        # extra backward): it must land before the buffer is cleared, and must not be mistaken for the next step's
        self._drain()
        self._done.clear()
        for _, p in self._externals():          # torch's zero_grad() clears every parameter the optimizer holds
            if p.grad is not None:
                p.grad.zero_()
        # the last step() cleared the buffer inside its kernel and NOTHING has asked for it since (FlatAgent.grad_dirty: a
        # backward into this agent during another agent's turn sets it): only then is the fill skipped
        if self._grad_is_zero and self.flat.attached() and not self.flat.grad_dirty:
            self._grad_is_zero = False      # whatever runs next writes into it again
            self.flat.ensure()
            return
        self._grad_is_zero = False
        self.flat.zero_grad()

    def set_grad_clip(self, grad_clip):
        """utils.clip_gradient() on a FlatAdam defers the clamp into the fused step kernel."""
        self._grad_clip = float(grad_clip)

    # run the bucketed exchange even in a process group of ONE rank (the sum over one rank is the identity): lets a single
    # GPU exercise the RCCL branch - communicator stream, three asynchronous handles in flight, wait ordering -
    # tests/test_gpu_rccl_world1.py.  Also settable from the environment (CIC_FORCE_GRAD_EXCHANGE=1).
    force_exchange = os.environ.get('CIC_FORCE_GRAD_EXCHANGE', '0') == '1'

    @classmethod
    def _distributed(cls):
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or cls.force_exchange)

    def begin_all_reduce(self, bucket=None):
        """Start the all-reduce of one bucket (default: every bucket not started yet) as soon as its gradient is final,
        so that it runs under the rest of the backward pass (the listener's gradient is complete before the speaker's
        backward starts, the speaker's logit layer before its BPTT loop).  At most once per bucket and step: a second
        backward before step() (gradient accumulation) must not start the exchange early."""
        if not self._distributed():
            return
        for name, (a, b) in self.buckets().items():
            if bucket is not None and name != bucket:
                continue
            if name in self._pending or name in self._done:
                raise RuntimeError(f'gradient bucket "{name}" was already exchanged in this step: begin_all_reduce() '
                                   f'runs once per bucket between zero_grad() and step()')
            self._pending[name] = dist.all_reduce(self.flat.grad[a:b], op=dist.ReduceOp.SUM, async_op=True)
            # RCCL queues its collectives on one stream in issue order; gloo (rehearsals on CPU or on a shared GPU) runs them
            # on worker threads, and several large exchanges in flight deadlocked it at 4 ranks (2 were fine): with gloo each
            # exchange completes before the next is issued.  The handle stays in _pending (waiting again is a no-op).
            if dist.get_backend() == 'gloo':
                self._pending[name].wait()

    def wait_pending(self):
        """Make the current stream wait for the exchanges of this step that are in flight (stream-ordered on RCCL: the host does
        not block).  The handles stay pending; step() waits again, which is then a no-op."""
        for h in self._pending.values():
            h.wait()

    def all_reduce_grads(self):
        """Every bucket summed over the ranks (one collective per bucket); returns the 1/world scale that is folded
        into the Adam kernel."""
        if not self._distributed():
            return 1.0
        for name in self.buckets():
            if name not in self._pending and name not in self._done:
                self.begin_all_reduce(name)
        for name, h in list(self._pending.items()):
            h.wait()                  # stream-ordered on the GPU backends: the host does not block
            self._done.add(name)
        self._pending.clear()
        return 1.0 / dist.get_world_size()

    def step(self):
        fl = self.flat
        g = self.param_groups[0]
        scale = self.all_reduce_grads() * self.grad_scale
        self._done.clear()
        if not any(p.requires_grad for p in fl.params):
            return
        fl.step += 1
        clip = self._grad_clip if self._grad_clip is not None else 3.0e38
        engine.clamp_adam(fl.flat, fl.grad, fl.exp_avg, fl.exp_avg_sq, g['lr'], fl.step, clip, g['betas'], g['eps'],
                          g['weight_decay'], scale, zero_grad=self.zero_grad_in_step)
        for name, p in self._externals():
            # torch's Adam skips a parameter without gradient (a frozen table: phase 2); the gradient is the owner's segment and
            # stays as it is (the owner's own step still needs it, and clears it if it clears)
            if not p.requires_grad or p.grad is None:
                continue
            if self.ext_owner is not None:
                self.ext_owner.all_reduce_grads()          # data parallel: the owner's bucket carries this gradient
            st = self._ext_state.get(name)
            if st is None:
                st = self._ext_state[name] = (torch.zeros(p.numel(), device=p.device), torch.zeros(p.numel(), device=p.device))
            engine.clamp_adam(p.data.view(-1), p.grad.view(-1), st[0], st[1], g['lr'], fl.step, clip, g['betas'], g['eps'],
                              g['weight_decay'], scale, zero_grad=False)
        self._grad_is_zero = bool(self.zero_grad_in_step)
        if self.zero_grad_in_step:
            fl.grad_dirty = False           # (a skipped update - status word set - leaves the run to raise; see status.py)

    # torch-compatible checkpoints: per-parameter state in parameter order
    def state_dict(self):
        fl = self.flat
        state = {}
        for i, (name, p, o) in enumerate(zip(fl.names, fl.params, fl.offsets)):
            n = p.numel()
            if o is None:                     # a shared table: this optimizer's own moments for it
                st = self._ext_state.get(name)
                if st is not None:
                    state[i] = dict(step=torch.tensor(float(fl.step)), exp_avg=st[0].view(p.shape).clone(),
                                    exp_avg_sq=st[1].view(p.shape).clone())
                continue
            state[i] = dict(step=torch.tensor(float(fl.step)),
                            exp_avg=fl.exp_avg[o:o + n].view(p.shape).clone(),
                            exp_avg_sq=fl.exp_avg_sq[o:o + n].view(p.shape).clone())
        g = self.param_groups[0]
        pg = dict(lr=g['lr'], betas=g['betas'], eps=g['eps'], weight_decay=g['weight_decay'], amsgrad=False,
                  params=list(range(len(fl.params))))
        return dict(state=state, param_groups=[pg])

    def load_state_dict(self, sd):
        fl = self.flat
        for i, (name, p, o) in enumerate(zip(fl.names, fl.params, fl.offsets)):
            st = sd['state'].get(i)
            if st is None:
                continue
            n = p.numel()
            if o is None:
                self._ext_state[name] = (st['exp_avg'].reshape(-1).to(p.device).float().clone(),
                                         st['exp_avg_sq'].reshape(-1).to(p.device).float().clone())
                fl.step = int(st['step'])
                continue
            fl.exp_avg[o:o + n].copy_(st['exp_avg'].reshape(-1))
            fl.exp_avg_sq[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            fl.step = int(st['step'])
        if sd.get('param_groups'):
            self.param_groups[0]['lr'] = sd['param_groups'][0].get('lr', self.param_groups[0]['lr'])


def load_optimizer_path(opt, curr_turn=None):
    """optimizer.py:9-22."""
    if opt.is_alternating:
        p = os.path.join(opt.start_from, curr_turn + '_optimizer.pth')
        return p if os.path.isfile(p) else None
    if opt.start_from is not None:
        return os.path.join(opt.start_from, 'optimizer.pth')
    return None


def define_optimizer(model, opt):
    """optimizer.py:25-27."""
    return FlatAdam(model, lr=opt.learning_rate, weight_decay=opt.weight_decay)


def load_state_dict(optimizer, optimizer_path, agent=''):
    """optimizer.py:30-40 (tensors only: weights_only=True)."""
    sd = torch.load(optimizer_path, map_location='cpu', weights_only=True)
    optimizer.load_state_dict(sd)
    print(f'\n Loaded {agent} optimizer from {optimizer_path} \n')
    return optimizer


def load_optimizer(model, opt):
    """optimizer.py:149-188.  In gumbel/multinomial alternating mode the listener turn is removed and
    both agents step every iteration (:90-95)."""
    start_from_exist = (vars(opt).get('start_from', None) is not None)
    optimizer_dict = {}
    if opt.is_alternating:
        for curr_turn in list(opt.alternating_turn):
            if curr_turn == 'speaker':
                o = define_optimizer(model.caption_generator, opt)
                if start_from_exist:
                    path = load_optimizer_path(opt, curr_turn)
                    if path:
                        o = load_state_dict(o, path, curr_turn)
                    elif not opt.share_embed and os.path.isfile(str(opt.speaker_stage_2_optimizer_path)):
                        o = load_state_dict(o, opt.speaker_stage_2_optimizer_path, curr_turn)
                else:
                    print('Loaded new "speaker" optimizer')
                optimizer_dict[curr_turn] = o
            elif curr_turn == 'listener':
                o = define_optimizer(model.vse, opt)
                if start_from_exist:
                    path = load_optimizer_path(opt, curr_turn)
                    if path:
                        o = load_state_dict(o, path, curr_turn)
                    elif not opt.share_embed and opt.initialize_retrieval:
                        p2 = os.path.join(os.path.split(opt.initialize_retrieval)[0], 'optimizer.pth')
                        if os.path.isfile(p2):
                            o = load_state_dict(o, p2, curr_turn)
                    else:
                        print('\n Using new "listener" optimizer \n')
                spk = optimizer_dict.get('speaker')
                if getattr(opt, 'share_embed', 0) and isinstance(spk, FlatAdam):
                    spk.ext_owner = o          # the shared table's gradient travels in the listener's bucket
                if opt.retrieval_reward == 'reinforce':
                    optimizer_dict[curr_turn] = o
                else:
                    optimizer_dict['speaker'] = {'speaker': optimizer_dict['speaker'], 'listener': o}
                    opt.alternating_turn.remove('listener')
    else:
        exist = load_optimizer_path(opt)
        if opt.phase == 1:
            o = define_optimizer(model.vse, opt)
        elif opt.phase in (2, 3):
            o = define_optimizer(model.caption_generator, opt)
        else:
            raise AssertionError(f'phase has to be 1,2 or 3 but got {opt.phase}')
        if start_from_exist and exist and os.path.isfile(exist):
            o = load_state_dict(o, exist)
        optimizer_dict['optimizer'] = o
    return optimizer_dict


def save_optimizer(opt, optimizer_dict):
    """optimizer.py:191-221."""
    def _save(o, name):
        path = os.path.join(opt.checkpoint_path, name)
        torch.save(o.state_dict(), path)
        print(f'\n optimizer saved to {path}')
    if opt.is_alternating:
        if opt.retrieval_reward == 'reinforce':
            for agent, o in optimizer_dict.items():
                _save(o, agent + '_optimizer.pth')
        else:
            for agent, o in optimizer_dict['speaker'].items():
                _save(o, agent + '_optimizer.pth')
    else:
        _save(optimizer_dict['optimizer'], 'optimizer.pth')


def fuse_zero_grad(optimizer_dict, on=True):
    """Let every FlatAdam clear its gradient buffer inside its clamp+Adam kernel (see FlatAdam.zero_grad_in_step)."""
    for v in optimizer_dict.values():
        for o in (v.values() if isinstance(v, dict) else [v]):
            if isinstance(o, FlatAdam):
                o.zero_grad_in_step = bool(on)


def accumulate_gradients(optimizer_dict, more_to_come):
    """Gradient accumulation under data parallelism: call with more_to_come=True before the backward of every micro-batch
    but the last, and with False before the last one.  While True the exchanges that overlap_gradient_exchange() starts
    from inside backward() stay off (they would sum a gradient that is not final, and a second start raises); the last
    backward starts them as usual and step() exchanges the rest.  Scale with FlatAdam.grad_scale = 1/n."""
    for v in optimizer_dict.values():
        for o in (v.values() if isinstance(v, dict) else [v]):
            if isinstance(o, FlatAdam):
                o.defer_exchange = bool(more_to_come)


def overlap_gradient_exchange(model, optimizer_dict):
    """Data-parallel runs: let the joint model start the listener's all-reduce from inside backward() (right after
    the listener's backward engines, before the speaker's), instead of after the whole backward pass.  With more than one
    backward per step (gradient accumulation) bracket the micro-batches with accumulate_gradients()."""
    lst = spk = None
    for v in optimizer_dict.values():
        for o in (v.values() if isinstance(v, dict) else [v]):
            if isinstance(o, FlatAdam) and o.module is getattr(model, 'vse', None):
                lst = o
            if isinstance(o, FlatAdam) and o.module is getattr(model, 'caption_generator', None):
                spk = o
    shared = bool(getattr(model, 'share_embed', 0))
    if lst is not None:
        # share_embed = 1: the speaker's backward still adds into the listener's gradient segment (the shared table), so the
        # listener's bucket leaves after the whole backward pass, and the speaker - whose step reads that segment before the
        # listener's step may clear it - is updated first (the reference's order, optimizer.py:233-239)
        lst._started_early = not shared
    model.listener_grads_ready = (lambda: None if lst.defer_exchange else lst.begin_all_reduce()) \
        if lst is not None and not shared else None
    # the speaker's logit bucket (final before the BPTT loop of its backward engine) leaves from inside backward() too
    if spk is not None and 'logit' in spk.buckets():
        model.speaker_logit_grads_ready = lambda: None if spk.defer_exchange else spk.begin_all_reduce('logit')
    else:
        model.speaker_logit_grads_ready = None
    # The speaker's BPTT loop is ONE launch that needs every CU (spk_bptt_seq_kernel): a collective that holds CUs would keep
    # part of its workgroups from becoming resident while the rest spin for them.  Its backward therefore lets the exchanges in
    # flight (the listener's bucket, started a few launches earlier) land first - a stream-side wait, no host block.
    cg = getattr(model, 'caption_generator', None)
    if cg is not None:
        agents = [o for o in (lst, spk) if o is not None]
        cg.exchange_barrier = lambda: [o.wait_pending() for o in agents]


def zeroing_optimizer(opt, optimizer_dict, optimizer):
    """optimizer.py:224-230."""
    if opt.retrieval_reward != 'reinforce' and opt.is_alternating:
        for agent in optimizer_dict['speaker'].keys():
            optimizer_dict['speaker'][agent].zero_grad()
    else:
        optimizer.zero_grad()


def update_optimizer(optimizer_dict, optimizer, opt):
    """optimizer.py:233-242: clamp then step, for one or both agents."""
    if opt.retrieval_reward != 'reinforce' and opt.is_alternating:
        agents = list(optimizer_dict['speaker'].values())
        for o in agents:                       # data-parallel: every exchange in flight before the first update waits
            if isinstance(o, FlatAdam) and o._distributed():
                for name in o.buckets():
                    if name not in o._pending and name not in o._done:
                        o.begin_all_reduce(name)
        # the agent whose exchange started first (the listener's, from inside backward) is updated first - a stable sort: without
        # an early start (single GPU, share_embed) the reference's order, speaker then listener, stands
        for o in sorted(agents, key=lambda o: 0 if getattr(o, '_started_early', False) else 1):
            utils.clip_gradient(o, opt.grad_clip)
            o.step()
    else:
        utils.clip_gradient(optimizer, opt.grad_clip)
        optimizer.step()
