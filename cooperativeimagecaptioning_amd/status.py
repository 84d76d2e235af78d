"""The sticky status word of the C ABI (include/cic.h, "status word") on the Python host.

Four recurrences of a step run as ONE launch each whose workgroups hand state to each other inside the launch; they need
every workgroup resident at once.  When a hand-off times out (another process holds CUs and CIC_SHARED_DEVICE was not set,
a CU-masked queue) the kernel ORs its bit into a caller-owned device word and the clamp+Adam kernels - which read the same
word - leave the weights alone.  This module owns that word (one int32 tensor per device), hands its address to every
engine call, and turns a set word into a CicError that names the loop and the step.  Nothing here synchronises the device:
the trainer copies the word to pinned memory beside the step's loss (train.LossLog), bench.py reads it once after its
timed region.
"""
import torch

from ._lib import CicError

GRU_FWD, GRU_BWD, TEACHER, BPTT, DECODE_STEP, UPDATE_SKIPPED = 1, 2, 4, 8, 16, 256
LOOPS = {
    GRU_FWD: "the listener's GRU pass (gru_seq_kernel)",
    GRU_BWD: "the listener's GRU BPTT loop (gru_seq_bwd_kernel)",
    TEACHER: "the speaker's teacher-forced recurrence (spk_teacher_seq_kernel)",
    BPTT: "the speaker's BPTT loop (spk_bptt_seq_kernel)",
    DECODE_STEP: "a sampling decode step's attention -> att2ctx + cell launch (attn_a2c_cell_kernel)",
}
_WORDS = {}


def word(device=None):
    """int32[4] on `device` (element 0 is the status word; the rest pads it to 16 bytes), zero when created."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    w = _WORDS.get(dev.index)
    if w is None:
        w = _WORDS[dev.index] = torch.zeros(4, dtype=torch.int32, device=dev)
    return w


def ptr(device=None):
    return word(device).data_ptr()


def describe(bits):
    names = [txt for b, txt in LOOPS.items() if bits & b]
    return '; '.join(names) if names else f'unknown bits {bits:#x}'


def raise_if_set(bits, where=''):
    """bits: the word's value as a Python int (already on the host)."""
    bits = int(bits)
    if bits == 0:
        return
    skipped = ' The optimiser update of that step and of every step after it was skipped: the weights are those of the ' \
              'last good step.' if bits & UPDATE_SKIPPED else ''
    raise CicError(
        f'a hand-off inside a one-launch recurrence timed out{(" (" + where + ")") if where else ""}: '
        f'{describe(bits)} (status word {bits:#x}).  Its workgroups were not all resident at the same time - something '
        f'else holds compute units of this GPU (a second process, a CU-masked queue).  Run processes that share a GPU '
        f'with CIC_SHARED_DEVICE=1 (per-step launches), then restart from the last checkpoint.' + skipped)


def check(device=None, where=''):
    """Synchronising read (tests, the end of a benchmark, checkpoint time)."""
    raise_if_set(int(word(device)[0].item()), where)


def clear(device=None):
    word(device).zero_()
