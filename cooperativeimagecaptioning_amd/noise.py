"""Random draws of the training step (dropout keep masks, Gumbel uniforms) from the library's
Philox4x32-10 counter RNG, or injected by the caller.

The reference draws Gumbel noise from torch's CPU generator and dropout masks from the device
generator (models/gumbel.py:6-11, nn.Dropout); neither stream can be reproduced on another
device, so parity tests inject the very arrays the reference/oracle used (``override``) and
production runs draw from (seed, call counter) so that a run is reproducible.
"""
import torch

from . import ops


class NoiseSource:
    def __init__(self, seed=0):
        self.seed = int(seed)
        self.counter = 0          # Philox offset in units of 2^32 calls: one fresh sub-stream per tensor
        self.override = None      # {tag: {att_keep,x_keep,out_keep,gumbel_u,pick}} numpy/torch arrays
        self._buf = {}            # persistent device buffers
        self._pending = []        # dropout masks requested but not yet drawn: (mask, p, Philox offset); see flush()

    def manual_seed(self, seed):
        self.seed, self.counter = int(seed), 0

    def flush(self):
        """Draw the dropout masks requested since the last flush (callers: right before the decode engine is launched)."""
        pend, self._pending = self._pending, []
        while pend:
            p = pend[0][1]
            chunk = [x for x in pend if x[1] == p][:8]          # one launch takes up to 8 masks of one keep probability
            pend = [x for x in pend if not any(x is c for c in chunk)]
            ops.dropout_keep_multi_([m for m, _, _ in chunk], p, self.seed, [o for _, _, o in chunk])

    def _next_offset(self):
        self.counter += 1
        return self.counter << 32

    def _get(self, key, shape, dtype, device):
        k = (key, tuple(shape), dtype, str(device))
        t = self._buf.get(k)
        if t is None:
            t = self._buf[k] = torch.empty(shape, dtype=dtype, device=device)
        return t

    def decode_noise(self, tag, B, K, H, E, V1, T, p, need_u, device, need_ss=False, need_ps=False, u_in_kernel=False):
        """u_in_kernel: the [T+1,B,V1] Gumbel uniforms are not materialised; out['u_stream'] = (seed, offset) names the
        Philox stream the decode kernels draw them from themselves - element for element the numbers ops.uniform_ would
        have written with that (seed, offset)."""
        if self.override is not None:
            ov = self.override.get(tag)
            if ov is None:
                raise KeyError(f'noise override has no entry for decode "{tag}"')
            out = {}
            for k, v in ov.items():
                if v is None:
                    continue
                t = torch.as_tensor(v)
                if k != 'att_keep' and t.shape[0] < T + 1:      # per-step arrays: pad to T+1 rows
                    pad = torch.ones((T + 1 - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)
                    t = torch.cat([t, pad], 0)
                if k.endswith('_keep'):
                    if p == 0.0:
                        continue
                    t = t.to(torch.uint8)
                elif k == 'pick':
                    t = t.long()
                else:
                    t = t.float()
                out[k] = t.to(device).contiguous()
            return out
        out = {}
        if p > 0.0:
            # the three masks of the decode in one launch, each from its own Philox sub-stream
            keys = (('att_keep', (B, K, H)), ('x_keep', (T + 1, B, E)), ('out_keep', (T + 1, B, H)))
            masks = [self._get((tag, key), shape, torch.uint8, device) for key, shape in keys]
            # drawn by flush(): the masks of ALL decodes of a launch (a decode pair: six) come from one kernel launch; each
            # keeps the Philox sub-stream it is assigned here, so the numbers do not depend on how launches are grouped
            self._pending += [(m, p, self._next_offset()) for m in masks]
            out.update({key: t for (key, _), t in zip(keys, masks)})
        if need_u and u_in_kernel:
            out['u_stream'] = (self.seed, self._next_offset())
        elif need_u:
            u = self._get((tag, 'gumbel_u'), (T + 1, B, V1), torch.float32, device)
            ops.uniform_(u, self.seed, self._next_offset())
            out['gumbel_u'] = u
        if need_ss:
            su = self._get((tag, 'ss_u'), (T + 1, B), torch.float32, device)
            ops.uniform_(su, self.seed, self._next_offset())
            out['ss_u'] = su
        if need_ps:
            pu = self._get((tag, 'ps_u'), (T + 1, B), torch.float32, device)
            ops.uniform_(pu, self.seed, self._next_offset())
            out['ps_u'] = pu
        return out
