"""Host -> HBM input pipeline for the training loop.

The reference's loader hands `get_batch(split)` dictionaries of host arrays (dataloader.py:171-245) and the trainer
copies them to the device at the top of every iteration (train.py:162-178): 37.7 MB per B = 128 batch, ~0.8 ms on
the critical path of a 5.8 ms step.  PrefetchLoader wraps ANY loader with that output contract and software-
pipelines the hand-over from the training thread itself:

    get_batch()   returns batch i, whose upload was issued one iteration earlier on a dedicated copy stream
                  (the compute stream is only made to WAIT for that upload's event);
    prefetch()    called by the trainer right after it has enqueued step i's kernels: pulls batch i+1 from the
                  wrapped loader, uploads it on the copy stream (the host is busy with the copy while the GPU
                  computes step i) and packs the reference captions for the CIDEr-D kernels.

No staging thread: the training thread is a tight loop of short GIL-holding launch calls, and a Python thread
beside it was measured to cost 10-27 ms per iteration in GIL hand-overs (tools/loader_bench.py).  What does run beside
it are the loader's reader threads inside numpy copies (no GIL): a loader with begin_batch() / end_batch()
(dataloader.DataLoader) is driven two batches ahead - batch i+2 is being copied into its pinned buffers while batch i+1
uploads and step i computes."""
import numpy as np
import torch

_TENSOR_KEYS = ('fc_feats', 'att_feats', 'att_masks', 'labels', 'masks')


class PrefetchLoader:
    """A loader that hands out PAGE-LOCKED arrays (dataloader.DataLoader, synthetic.SyntheticLoader(pin=True)) is
    uploaded with real asynchronous DMAs: the host only enqueues them.  Arrays in pageable memory are uploaded as
    they are: `.to(device, non_blocking=True)` is then a synchronous copy staged by the runtime, which holds the
    host for the 37.7 MB (measured, tools/loader_bench.py: 5.5 ms / iteration against 4.9 from pinned memory; an
    explicit pageable -> pinned staging copy in this class was far slower still, 33 ms) - so pin at the source."""

    def __init__(self, loader, device, split='train'):
        self.loader, self.device, self.split = loader, torch.device(device), split
        self.vocab_size = getattr(loader, 'vocab_size', None)
        self.seq_length = getattr(loader, 'seq_length', None)
        self._stream = torch.cuda.Stream(device=self.device)
        self._next = None
        self._assembling = None   # begin_batch() handle of the batch after the uploaded one (two-phase loaders)
        self._uploads = []        # completion events of the last two uploads
        self._two_phase = hasattr(loader, 'begin_batch') and hasattr(loader, 'end_batch')
        self.pageable_bytes = 0   # bytes uploaded from pageable memory (0 with a pinning loader, but for the packed captions)

    def ahead(self):
        """Batches pulled from the wrapped loader but not handed to the trainer yet (a resume replays them)."""
        return (0 if self._next is None else 1) + (0 if self._assembling is None else 1)

    def state_dict(self):
        """The wrapped loader's position such that a resume hands out the oldest batch the trainer has NOT consumed yet
        (pulled ahead by this class): the snapshot that batch's begin_batch() took, across an epoch wrap too."""
        inner = self.loader
        if not hasattr(inner, 'state_dict'):
            return None
        if self._next is not None and self._next[0].get('_loader_state') is not None:
            return inner.state_dict(snapshot=self._next[0]['_loader_state'])
        if self._next is None and self._assembling is not None and self._assembling.get('state') is not None:
            return inner.state_dict(snapshot=self._assembling['state'])
        return inner.state_dict(rewind=self.ahead()) if self.ahead() else inner.state_dict()

    def _pinned(self, key, t):
        if not t.is_pinned():
            self.pageable_bytes += t.numel() * t.element_size()
        return t

    def prefetch(self):
        """Pull the next batch and upload it on the copy stream (call after the current step is enqueued)."""
        if self._next is not None:
            return
        if self._two_phase:
            handle = self._assembling if self._assembling is not None else self.loader.begin_batch(self.split)
            data = self.loader.end_batch(handle)                   # copied while the previous step was computing
            data['_loader_state'] = handle.get('state')
            # the loader rotates three sets of pinned buffers: the set the next assembly writes was the source of the
            # upload issued two calls ago - that DMA has long finished, but nothing may overwrite it before it has
            if len(self._uploads) >= 2:
                self._uploads[-2].synchronize()
            self._assembling = self.loader.begin_batch(self.split)  # the batch after it: its copies start now
        else:
            data = self.loader.get_batch(self.split)
        out = dict(data)
        with torch.cuda.stream(self._stream):
            for k in _TENSOR_KEYS:
                v = data.get(k)
                if v is not None:
                    out[k] = self._pinned(k, torch.as_tensor(v)).to(self.device, non_blocking=True)
            gts = data.get('gts')
            if gts is not None and len(gts) and len(gts[0]):
                off = np.zeros(len(gts) + 1, np.int32)
                off[1:] = np.cumsum([len(g) for g in gts])
                refs = np.ascontiguousarray(np.concatenate([np.asarray(g) for g in gts], 0).astype(np.int32))
                ref_off = self._pinned('ref_off', torch.from_numpy(off)).to(self.device, non_blocking=True)
                ref_off.max_refs = int(max(len(g) for g in gts))     # as engine.pack_refs: lets the reward skip a fallback launch
                out['_cic_refs'] = (gts, self._pinned('refs', torch.from_numpy(refs)).to(self.device, non_blocking=True),
                                    ref_off)   # see AlternatingJointModel._refs
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._uploads = (self._uploads + [ev])[-2:]
        self._next = (out, ev)

    def get_batch(self, split=None):
        if self._next is None:
            self.prefetch()
        data, ev = self._next
        self._next = None
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)                                       # device-side wait: the host does not block
        for v in data.values():                                  # allocated on the copy stream, consumed on this one
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)
        if data.get('_cic_refs') is not None:
            for v in data['_cic_refs'][1:]:
                v.record_stream(cur)
        return data

    def close(self):
        if self._assembling is not None:
            self.loader.end_batch(self._assembling)                # let the reader threads finish with the pinned buffers
        self._next = self._assembling = None
