#!/usr/bin/env python3
"""Headline benchmark: joint-step images/sec of the AlternatingJointModel speaker<->listener
training step (att2in2 speaker + VSE-fc listener, straight-through Gumbel tau=1, CIDEr-D
self-critical term, B=128 per GPU, seq_len 16, 36x2048 region features, vocab 9487).

One step = zero_grad -> forward (sampled decode, listener on the generated captions, greedy
decode, CIDEr-D reward, loss) -> backward -> [RCCL all-reduce of the two flat gradient buffers]
-> fused clamp+Adam for both agents, with the batch already resident in HBM.

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0).  `roofline` is the per-timestep attention kernel (HBM-bound):
algorithmic bytes per launch (151,696 B x the 2B images one launch of the paired decodes covers) /
average duration of the in-step launches (HIP events of a caller-owned cic_timer on the step's stream,
collected over a few steps after the timed region) / 8 TB/s; `roofline_mfma` is the logit product
(2 x 2B x 512 x 9488 flop per launch / in-step duration / 157.3 TF/s).  `cpu_baseline` is the CPU
oracle (oracle/, a restatement of the reference pinned by golden vectors) timed on this host
for a bounded number of steps of the same workload.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ATTN_BYTES_PER_IMAGE = 151696          # p_att + att rows + att_h + att_res + alpha, f32
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
MFMA_F32_PEAK_TF = 157.3               # MI355X_MICROARCH.md: f32-input MFMA, 157.3 TFLOP/s (spec)
MFMA_BF16_PEAK_TF = 2500.0             # MI355X_MICROARCH.md: bf16 MFMA, ~2.5 PFLOP/s dense
# MI355X_MICROARCH.md, "Indexed rows: gather into LDS": a 38 MB table that stays in the Infinity Cache is read at 8.6 TB/s
# chip-wide.  The attention's working set (att + p_att of 2B images = 37.7 MB) is such a table: re-read every decode step, it
# is served by the Infinity Cache, not by HBM - `frac_of_cache_level` prices the kernel against THAT level beside the HBM `frac`
MALL_38MB_GBS = 8600.0


def host_cores():
    """(cores this process may run on, physical cores of the host, logical CPUs of the host)."""
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    logical = os.cpu_count() or usable
    physical = None
    try:
        seen = set()
        phys_id = core_id = None
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.startswith('physical id'):
                    phys_id = ln.split(':')[1].strip()
                elif ln.startswith('core id'):
                    core_id = ln.split(':')[1].strip()
                elif not ln.strip():
                    if phys_id is not None and core_id is not None:
                        seen.add((phys_id, core_id))
                    phys_id = core_id = None
        physical = len(seen) or None
    except OSError:
        pass
    return usable, physical, logical


def cpu_baseline(opt, steps_budget_s=20.0):
    """The oracle's joint step (fwd + bwd + clamp + Adam), same workload, on the host cores."""
    from oracle import joint as J
    from cooperativeimagecaptioning_amd import synthetic
    import numpy as np
    # BASELINE.md §3 asks for the host's cores.  A GPU box hands ONE GPU's share of its host to a job (16 cores on this pool);
    # the affinity mask is the authority where it is narrower, and the threads never exceed the physical cores (SMT siblings
    # slow the oracle's GEMMs down).  The cap and the host's real core counts are stated in `sample`.
    usable, physical, logical = host_cores()
    cap = 16
    nthreads = max(1, min(cap, usable, physical or usable))
    torch.set_num_threads(nthreads)
    torch.manual_seed(0)
    B = opt.batch_size
    batch = synthetic.make_batch(opt, seed=1234)
    g = torch.Generator().manual_seed(0)

    def lin(o, i):
        r = 1.0 / np.sqrt(i)
        return ((torch.rand(o, i, generator=g) * 2 - 1) * r).requires_grad_(True), \
               ((torch.rand(o, generator=g) * 2 - 1) * r).requires_grad_(True)
    H, E, A, D, V, Jd = opt.rnn_size, opt.input_encoding_size, opt.att_hid_size, opt.att_feat_size, opt.vocab_size, opt.vse_embed_size
    Ps = {'embed.0.weight': torch.randn(V + 2, E, generator=g).requires_grad_(True)}
    for nm, (o, i) in {'att_embed.0': (H, D), 'logit': (V + 1, H), 'ctx2att': (A, H), 'core.a2c': (2 * H, H),
                       'core.i2h': (5 * H, E), 'core.h2h': (5 * H, H), 'core.attention.h2att': (A, H),
                       'core.attention.alpha_net': (1, A)}.items():
        Ps[nm + '.weight'], Ps[nm + '.bias'] = lin(o, i)
    Pl = {'txt_enc.embed.weight': ((torch.rand(V + 2, E, generator=g) - .5) * .2).requires_grad_(True)}
    Pl['img_enc.fc.weight'], Pl['img_enc.fc.bias'] = lin(Jd, opt.fc_feat_size)
    Pl['txt_enc.rnn.weight_ih_l0'], Pl['txt_enc.rnn.bias_ih_l0'] = lin(3 * Jd, E)
    Pl['txt_enc.rnn.weight_hh_l0'], Pl['txt_enc.rnn.bias_hh_l0'] = lin(3 * Jd, Jd)
    cfg = {k: getattr(opt, k) for k in vars(opt)}
    st_s, st_l = {}, {}
    p = opt.drop_prob_lm
    T = opt.seq_length
    times = []
    t_all = time.time()
    n = 0
    while True:
        rs = torch.Generator().manual_seed(100 + n)

        def keeps():
            return dict(att_keep=(torch.rand(B, 36, H, generator=rs) >= p).float(),
                        x_keep=(torch.rand(T + 1, B, E, generator=rs) >= p).float(),
                        out_keep=(torch.rand(T + 1, B, H, generator=rs) >= p).float())
        t0 = time.time()
        noise = {'sample': keeps(), 'greedy': keeps()}      # Gumbel uniforms are drawn inside (torch.rand)
        for P in (Ps, Pl):
            for v in P.values():
                v.grad = None
        loss, _ = J.joint_forward(Ps, Pl, cfg, batch, noise, 'speaker', True)
        loss.backward()
        with torch.no_grad():
            J.clamp_adam_step({k: v for k, v in Ps.items()}, {k: v.grad for k, v in Ps.items()}, st_s, opt.learning_rate, opt.grad_clip)
            J.clamp_adam_step({k: v for k, v in Pl.items()}, {k: v.grad for k, v in Pl.items()}, st_l, opt.learning_rate, opt.grad_clip)
        times.append(time.time() - t0)
        n += 1
        if n >= 2 and (time.time() - t_all > steps_budget_s or n >= 12):
            break
    steady = sorted(times[1:])[len(times[1:]) // 2]
    return dict(value=B / steady, unit='images/s', cores=torch.get_num_threads(), kind='port',
                host_physical_cores=physical, host_logical_cpus=logical, cores_usable_by_this_process=usable,
                sample=f'{n} joint steps (first discarded, median of the rest) of the same B={B} gumbel+CIDEr-D '
                       f'workload on the CPU oracle, {steady * 1e3:.0f} ms/step; {torch.get_num_threads()} threads = '
                       f'min(cap {cap}: one GPU\'s share of the host, {usable} CPUs in the affinity mask, '
                       f'{physical} physical cores of the host; {logical} logical CPUs)')


def attention_launch_time(model, batch, stream, iters=200):
    """Average duration of one attention launch of the step, HIP events on the step's stream (cic_attn_fwd_timed).
    In the step the sampled and the greedy decode advance together, so ONE launch covers 2B images (B workgroups
    per decode): the timed launches have that geometry, on `att` tensors of the step's size, interleaved with a
    kernel that streams the ~70 MB the other kernels of a paired decode step move between two attention launches
    (weights, logits, noise), so that att / p_att are in the cache state they have inside the step."""
    import ctypes as C
    from cooperativeimagecaptioning_amd import _lib
    cg = model.caption_generator
    B, K, H = 2 * batch['att_feats'].shape[0], batch['att_feats'].shape[1], cg.rnn_size
    dev = batch['att_feats'].device
    att = torch.randn(B, K, H, device=dev).abs_()
    p_att = torch.randn(B, K, H, device=dev)
    att_h = torch.randn(B, H, device=dev)
    w = cg.core.attention.alpha_net.weight.data.view(-1)
    ba = cg.core.attention.alpha_net.bias.data
    res, al = torch.empty(B, H, device=dev), torch.empty(B, K, device=dev)
    fn = _lib.lib.cic_attn_fwd_timed
    fn.argtypes = [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.c_void_p]
    pollute = torch.randn(70 * (1 << 20) // 4, device=dev)
    us, us_warm = C.c_double(0.0), C.c_double(0.0)
    _lib.check(fn(att_h.data_ptr(), p_att.data_ptr(), att.data_ptr(), w.data_ptr(), ba.data_ptr(), res.data_ptr(),
                  al.data_ptr(), B, K, H, H, iters, pollute.data_ptr(), pollute.numel(), C.byref(us),
                  stream.cuda_stream), 'cic_attn_fwd_timed')
    _lib.check(fn(att_h.data_ptr(), p_att.data_ptr(), att.data_ptr(), w.data_ptr(), ba.data_ptr(), res.data_ptr(),
                  al.data_ptr(), B, K, H, H, iters, None, 0, C.byref(us_warm), stream.cuda_stream), 'cic_attn_fwd_timed')
    return {'attn_fwd': dict(ms=us.value * iters / 1e3, n=iters, warm_us=us_warm.value, images=B)}


def _profile_doc(names):
    for n in names:
        try:
            with open(os.path.join(ROOT, 'profiles', n)) as f:
                return json.load(f)
        except OSError:
            continue
    return None


def pmc_traffic(images):
    """HBM-side bytes per attention launch from the committed PMC summary (profiles/, separate FETCH_SIZE and
    WRITE_SIZE passes of this same command, tools/pmc_summary.py), for the launch geometry of the step."""
    doc = _profile_doc(['r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json'])
    for prefix in ('attn_a2c_cell_kernel', 'attn_fwd_cols_kernel'):      # (r4: the fused launch; before: the attention alone)
        for k in (doc or {}).get('kernels', []):
            if k['kernel'].startswith(prefix) and k['grid_threads'] == images * 1024:
                return k['total_bytes']
    return None


def pmc_mfma_util():
    """Counter-derived MFMA utilisation of the logit walker from the committed PMC pass (profiles/r03_pmc_mfma.json,
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ... of this same command), or None before that pass exists."""
    doc = _profile_doc(['r04_pmc_mfma.json', 'r03_pmc_mfma.json', 'r02_pmc_mfma.json'])
    for k in (doc or {}).get('kernels', []):
        if k['kernel'].startswith('gemm_ldsb2'):
            return k.get('mfma_util')
    return None


def attention_phase_stamps(live):
    """Duration of the attention phase INSIDE attn_a2c_cell_kernel (attention -> in-launch hand-off -> att2ctx + cell; the
    attention is no launch of its own any more), from s_memrealtime stamps of the development build: tools/attn_phase_stamps.py
    run as a child process on this GPU (live), else the summary committed under profiles/.  -> (dict or None, source)."""
    import subprocess
    if live:
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'attn_phase_stamps.py')], capture_output=True,
                               text=True, timeout=240)
            for ln in reversed(r.stdout.strip().splitlines()):
                if ln.startswith('{'):
                    d = json.loads(ln)
                    if d.get('phase_span_us'):
                        return d, 'live: tools/attn_phase_stamps.py (development build) as a child process of this run'
        except Exception as e:                  # reporting only: never lose the measured line over it
            print(f'bench.py: live attention-phase stamps failed ({type(e).__name__}: {e}); using profiles/', file=sys.stderr)
    for name in ('r04_attn_phase_stamps.json',):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                d = json.load(f)
            if d.get('phase_span_us'):
                return d, 'profiles/' + name
        except (OSError, ValueError):
            continue
    return None, None


def trace_kernel_us(prefix):
    """In-step average duration (us) of the kernel whose name starts with `prefix` in the committed rocprofv3 kernel trace of
    this command (profiles/r03_step_breakdown.md, written by tools/trace_summary.py from `rocprofv3 --kernel-trace`), or None."""
    for name in ('r04_step_breakdown.md', 'r03_step_breakdown.md', 'r03a_step_breakdown.md'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                for ln in f:
                    cells = [c.strip() for c in ln.split('|')]
                    if len(cells) > 5 and cells[1].strip('`').startswith(prefix):
                        return float(cells[4])
        except (OSError, ValueError):
            continue
    return None


def _rccl_version():
    try:
        return '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:                      # reporting only: never lose the measured line over it
        return f'unknown ({type(e).__name__})'


def launch_ranks(n, argv):
    """`bench.py --gpus N` started directly: run the N ranks under torch.distributed.run as a CHILD process (the parent has
    not touched the GPU and never does), pass its stdout through (rank 0 prints the one JSON line) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--batches', type=int, default=4, help='distinct synthetic batches resident in HBM, served round robin')
    ap.add_argument('--profile-steps', type=int, default=5, help='extra steps after the timed region with in-step kernel timing')
    ap.add_argument('--no-phase-stamps', action='store_true', help='attention-phase stamps from profiles/ instead of a live child process')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N > 1: 'nccl' (= RCCL over xGMI, the "
                    "default) or 'gloo' (rehearsal of the multi-process path on a box with fewer GPUs than ranks)")
    ap.add_argument('--same-device', action='store_true', help='rehearsal: every rank uses cuda:0 (gloo backend only)')
    ap.add_argument('--compute-dtype', default='f32', choices=['f32', 'bf16'],
                    help="'f32' (default: the reference's arithmetic - THE bench line) or 'bf16': the reduced-precision variant "
                         "(profiling only: its line says dtype bf16 and is not the headline)")
    args = ap.parse_args()

    if os.environ.get('CIC_HANG_DUMP'):          # diagnostics: Python stacks of every thread after N seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['CIC_HANG_DUMP']), exit=True)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process makes NO GPU call; it starts the N ranks as a child
        # (`python -m torch.distributed.run ... bench.py <same flags>`), relays rank 0's JSON line and the exit code
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.same_device:
            assert args.backend == 'gloo', '--same-device is a gloo rehearsal (RCCL wants one GPU per rank)'
            os.environ['CIC_SHARED_DEVICE'] = '1'     # several ranks compute on one GPU: no launches that need the whole chip
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group(args.backend)  # 'nccl' = RCCL over xGMI
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local_rank)

    from cooperativeimagecaptioning_amd import build as _build
    # a checkout without the (git-ignored) library: rank 0 compiles it in-tree (build() links to a temporary name and
    # renames it into place); EVERY rank then passes the same barrier, whatever it saw on disk, before it loads it
    if rank == 0 and _build.needs_build():
        with contextlib.redirect_stdout(sys.stderr):
            _build.build(force=False)
    if world > 1:
        dist.barrier()
    from cooperativeimagecaptioning_amd import models, optimizer as optim, synthetic, engine, status
    from cooperativeimagecaptioning_amd.misc import rewards

    opt = synthetic.default_opt(batch_size=args.batch, compute_dtype=args.compute_dtype)
    torch.manual_seed(0)                       # identical initial weights on every rank
    rewards.init_scorer('corpus')
    model = models.AlternatingJointModel(opt).to(dev).train()
    model.caption_generator.noise.manual_seed(1000 + rank)
    with contextlib.redirect_stdout(sys.stderr):        # the reference-style progress prints stay off stdout: ONE JSON line there
        optimizer_dict = optim.load_optimizer(model, opt)
    if world > 1:
        optim.overlap_gradient_exchange(model, optimizer_dict)   # listener all-reduce under the speaker backward
    optim.fuse_zero_grad(optimizer_dict)        # the gradient buffers are cleared inside the clamp+Adam kernels
    # per-rank shard of the global batch; `--batches` distinct batches stay resident in HBM and are served round robin
    # (their reference captions are packed for the CIDEr-D kernels on first use: host work outside the metric, like
    # the loader's; the n-gram / document-frequency / score kernels run in every step)
    batches = [synthetic.make_batch(opt, seed=1234 + rank + 1000 * i, device=dev) for i in range(max(1, args.batches))]
    batch = batches[0]
    turn = opt.alternating_turn[0]
    optimizer = optimizer_dict[turn]
    counter = [0]

    def step():
        b = batches[counter[0] % len(batches)]
        counter[0] += 1
        optim.zeroing_optimizer(opt, optimizer_dict, optimizer)
        loss = model(b['fc_feats'], b['labels'], b['masks'], b, b['att_feats'], b['att_masks'],
                     is_alternating=True, alternating_turn=turn)
        loss.backward()
        optim.update_optimizer(optimizer_dict, optimizer, opt)
        return loss

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    stream = torch.cuda.current_stream(dev)       # the engines launch on torch's current stream
    for _ in range(args.warmup):
        loss = step()
    barrier()
    # K timed steps; an event at every step boundary (same stream, no host sync) gives the per-step times for the median
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    host_t = []
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
        host_t.append(time.perf_counter() - t0)          # when the host had ENQUEUED step i
    barrier()
    dt = time.perf_counter() - t0
    # a hand-off of a one-launch recurrence that timed out (status.py) ends the benchmark with a non-zero exit code on the rank
    # that saw it: the optimiser kernels skipped their updates from that step on, and a line measured over skipped work is void
    status.check(dev, f'bench.py rank {rank}, during the {args.warmup} warm-up + {args.steps} timed steps')
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2]
    # how far the host runs ahead of the device: (device time at which step i ended) - (host time at which it was enqueued).
    # Growing from step to step = the device sets the pace (the host waits in the end); ~0 throughout = launch-bound.
    ahead = [marks[0].elapsed_time(marks[i + 1]) - host_t[i] * 1e3 for i in range(args.steps)]
    host_ms = (host_t[-1] - host_t[0]) / max(args.steps - 1, 1) * 1e3
    # roofline legs, AFTER the timed region: the same step a few more times with a cic_timer attached to the decodes, which
    # brackets every in-step launch of the attention / logit / sampler kernels with HIP events on this stream
    # (every rank runs them - the steps contain the gradient exchange -, rank 0 reads its timer)
    prof = {}
    if args.profile_steps > 0:
        timer = engine.KernelTimer()
        model.caption_generator.timer = timer
        for _ in range(args.profile_steps):
            loss = step()
        barrier()
        status.check(dev, f'bench.py rank {rank}, during the profiled steps')
        model.caption_generator.timer = None
        if rank == 0:
            prof = timer.collect()
            prof['bracket_overhead_us'] = timer.bracket_overhead_us()
            prof['attn_microbench'] = attention_launch_time(model, batch, stream)['attn_fwd']
            # (no child process under a profiler - its preloaded library would trace the child into the same output -, nor
            # next to other ranks)
            profiled = any(k in os.environ.get('LD_PRELOAD', '') for k in ('rocprof', 'roctracer')) or 'ROCPROF' in ''.join(os.environ)
            prof['attn_phase'] = attention_phase_stamps(live=(world == 1 and not profiled and not args.no_phase_stamps))
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    final_loss = float(loss.detach())
    assert final_loss == final_loss, 'loss is NaN'

    if rank == 0:
        B = args.batch
        n_img = 2 * B                                      # images per attention launch: the paired decodes of the step
        attn = prof.get('attn_fwd', dict(ms=0.0, n=0))
        ovh = prof.get('bracket_overhead_us', 0.0)        # what an event pair adds by itself (measured, same stream)
        attn_raw_us = attn['ms'] * 1e3 / max(attn['n'], 1)
        micro = prof.get('attn_microbench', {})
        micro_us = micro.get('ms', 0.0) * 1e3 / max(micro.get('n', 1), 1)
        # A 9 us kernel is shorter than what a HIP event pair costs by itself (~4.6 us for an empty pair): the in-step
        # brackets bound its duration (raw: too long, minus an empty pair: too short) but do not resolve it.  `achieved`
        # uses the duration of launches of the step's geometry in the step's cache state (cic_attn_fwd_timed: HIP events
        # over 200 launches interleaved with a kernel that streams what a decode step moves between two attention
        # launches, that kernel's own time subtracted) - the figure the rocprofv3 trace of the in-step launches agrees with.
        # (r4) In the step the attention is the first PHASE of attn_a2c_cell_kernel (attention -> in-launch hand-off -> att2ctx
        # product + cell: one launch per decode step).  Its duration comes from s_memrealtime stamps of the development build:
        # earliest wave start -> last att_res store issued, over all workgroups of a launch (the chip-wide span the launch's
        # algorithmic bytes are divided by).  The stand-alone attention launch (the events above) and the whole fused kernel
        # (in-step brackets; rocprofv3 trace) are reported beside it.
        phase, phase_src = prof.get('attn_phase', (None, None))
        attn_us = phase['phase_span_us'] if phase else micro_us
        achieved = (ATTN_BYTES_PER_IMAGE * n_img) / (attn_us * 1e-6) / 1e9 if attn_us > 0 else 0.0
        fused_bracket_us = max(attn_raw_us - ovh, 0.0)
        fused_trace_us = trace_kernel_us('attn_a2c_cell_kernel')
        fused_us = max(fused_bracket_us, fused_trace_us or 0.0)
        H_ = opt.rnn_size
        # second phase, per launch: a2c weights once, the five gate pre-activations read + the two candidates written back, c, h, c',
        # out, att_res written and read, the keep bytes
        cell_bytes = 2 * H_ * H_ * 4 + n_img * H_ * 4 * (5 + 2 + 1 + 3 + 2) + n_img * H_
        fused_bytes = ATTN_BYTES_PER_IMAGE * n_img + cell_bytes
        fused_gbs = fused_bytes / (fused_us * 1e-6) / 1e9 if fused_us > 0 else 0.0
        lg = prof.get('logit_gemm', dict(ms=0.0, n=0))
        lg_bracket_us = max(lg['ms'] * 1e3 / max(lg['n'], 1) - ovh, 0.0)
        # The in-step bracket minus an empty event pair UNDER-reads a ~29 us kernel by a few percent (round 2: 27.8 us against
        # 30.1 us in the rocprofv3 trace of the same command).  The line therefore prices the kernel at the LONGER of the two
        # durations: the live bracket, and the in-step average of the committed trace (profiles/, same build).
        lg_trace_us = trace_kernel_us('gemm_ldsb2bf_walk_kernel')
        lg_us = max(lg_bracket_us, lg_trace_us or 0.0)
        lg_flop = 2.0 * n_img * opt.rnn_size * (opt.vocab_size + 1)      # one launch: [2B, 512] x [512, 9488]
        lg_tf = lg_flop / (lg_us * 1e-6) / 1e12 if lg_us > 0 else 0.0
        out = {
            'metric': 'joint-step images/sec (B=128, seq16)', 'value': B * world * args.steps / dt, 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'median_ms_per_step': median_ms,
            'host_enqueue_ms_per_step': host_ms, 'device_behind_host_ms': {'first': ahead[0], 'median': sorted(ahead)[len(ahead) // 2],
                                                                          'last': ahead[-1]},
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.compute_dtype, 'data': 'synthetic',
            'config': {'workload': 'AlternatingJointModel joint step, att2in2 speaker + VSE-fc listener, ST-Gumbel '
                                   'tau=1 + self-critical CIDEr-D, 36x2048 att_feats, vocab 9487, seq_len 16, '
                                   'dropout 0.5, clamp 0.1 + Adam both agents (BASELINE configs[2])',
                       'batch_per_gpu': B, 'global_batch': B * world, 'parallelism': f'dp{world}',
                       'paired_decodes': bool(getattr(model.caption_generator, 'last_pair_fused', False)), 'resident_batches': len(batches),
                       'excluded': 'host packing + upload of the reference captions (once per resident batch, the loader\'s job)',
                       'final_loss': final_loss,
                       'world': world, 'backend': (args.backend if world > 1 else None),
                       'rccl': _rccl_version() if world > 1 and args.backend == 'nccl' else None,
                       'gradient_buckets': {a: list(o.buckets()) for a, o in optimizer_dict.get('speaker', {}).items()}
                       if isinstance(optimizer_dict.get('speaker'), dict) else None},
            # HBM-bound kernel of the path: the per-timestep attention.  achieved = algorithmic bytes of one launch / the
            # average duration of the in-step launches (HIP events of a cic_timer on the step's stream)
            'roofline': {'bound': 'hbm',
                         'kernel': 'attn_a2c_cell_kernel<5, float>, attention phase (per-timestep top-down attention of 2B rows; the '
                                   'att2ctx product + cell of the step follow in the same launch behind an in-launch hand-off)',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         # the level the kernel actually reads from: its 37.7 MB working set is Infinity-Cache resident
                         'frac_of_cache_level': achieved / MALL_38MB_GBS, 'cache_level_peak': MALL_38MB_GBS,
                         'cache_level': 'Infinity Cache, 38 MB gathered table: 8.6 TB/s chip-wide (MI355X_MICROARCH.md)',
                         'traffic': pmc_traffic(n_img), 'avg_launch_us': attn_us,
                         'launches_timed': (phase or {}).get('launches_stamped', 0),
                         'timing': 'attention phase of the in-step launches: s_memrealtime stamps of the development build, earliest '
                                   'wave start -> last att_res store issued over all 256 workgroups of a launch (median over launches)'
                                   if phase else 'stand-alone attention launches (HIP events); no phase stamps available',
                         'phase_source': phase_src, 'phase_per_workgroup_median_us': (phase or {}).get('per_workgroup_median_us'),
                         # the whole fused launch: HIP-event brackets around every in-step launch (cic_timer) minus an empty pair, and
                         # the committed rocprofv3 trace; priced at the longer one against the bytes of BOTH phases
                         'whole_kernel': {'avg_launch_us': fused_us, 'avg_launch_us_bracket': fused_bracket_us,
                                          'avg_launch_us_trace': fused_trace_us, 'launches_bracketed': attn['n'],
                                          'algorithmic_bytes_per_launch': fused_bytes, 'achieved': fused_gbs,
                                          'frac': fused_gbs / HBM_PEAK_GBS},
                         # the attention as a launch of its own (the form a process on a shared GPU runs), HIP events over 200
                         # launches of the step's geometry interleaved with a cache-polluting kernel (its time subtracted)
                         'standalone_launch_us': micro_us, 'standalone_launches_timed': micro.get('n', 0),
                         'standalone_launch_us_l2_warm': micro.get('warm_us'),
                         'in_step_bracket_us': attn_raw_us, 'empty_bracket_us': ovh,
                         'images_per_launch': n_img, 'algorithmic_bytes_per_launch': ATTN_BYTES_PER_IMAGE * n_img,
                         'traffic_covers': 'the whole fused launch (both phases): compare with whole_kernel.algorithmic_bytes_per_launch',
                         'traffic_source': 'profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/pmc_summary.py)'},
            # MFMA-bound kernel of the path: the hidden -> vocabulary logit product (with its fused log-softmax / sampler
            # partials), exact-f32 MFMA
            'roofline_mfma': {'bound': 'mfma', 'kernel': 'gemm_ldsb2bf_walk_kernel<32, 1> (logit product [2B,512]x[512,9488] + row partials; f32 results from six '
                                        'bf16-part products per k on v_mfma_f32_16x16x32_bf16, priced against the f32 MFMA peak)',
                              'achieved': lg_tf, 'peak': MFMA_F32_PEAK_TF, 'unit': 'TFLOP/s', 'frac': lg_tf / MFMA_F32_PEAK_TF,
                              'flop_per_launch': lg_flop, 'avg_launch_us': lg_us, 'launches_timed': lg['n'],
                              'avg_launch_us_bracket': lg_bracket_us, 'avg_launch_us_trace': lg_trace_us,
                              'timing': 'the longer of (a) HIP event brackets around every in-step launch (cic_timer) minus an empty '
                                        'event pair and (b) the in-step average of the committed rocprofv3 trace (profiles/)',
                              # the kernel issues SIX bf16 part products per k on the bf16 pipe (2.5 PFLOP/s dense): the share
                              # of THAT pipe's peak it keeps busy, by arithmetic and by counter
                              'frac_of_issued_pipe': (6.0 * lg_tf) / MFMA_BF16_PEAK_TF,
                              'issued_pipe_peak_tflops': MFMA_BF16_PEAK_TF,
                              'mfma_util_pmc': pmc_mfma_util()},
        }
        for k in ('sampler', 'attn_bwd'):
            v = prof.get(k)
            if v and v['n']:
                out.setdefault('kernel_us', {})[k] = max(v['ms'] * 1e3 / v['n'] - ovh, 0.0)
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.synchronize()
            out['cpu_baseline'] = cpu_baseline(opt, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
