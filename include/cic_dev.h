/* cic_dev.h - development build of the library ONLY (libcic_hip_dev.so = the same sources compiled with
 * -DCIC_DEVTOOLS).  The product library libcic_hip.so exports none of these and holds no mutable global state: its
 * dispatch choices are compile-time constants and its kernels carry no stamp code.  tools/ load the development
 * build for A/B timing of kernel variants and for in-kernel phase stamps (tools/_devlib.py). */
#ifndef CIC_DEV_H
#define CIC_DEV_H
#include "cic.h"
#ifdef __cplusplus
extern "C" {
#endif

/* diagnostics: shader clock in MHz (out2[0]) measured over `spin` dependent FMAs; an empty launch */
int cic_debug_clock_mhz(float* out2, int spin, cic_stream_t s);
int cic_debug_empty(int grid, int block, cic_stream_t s);
/* diagnostics: fill-rate probe.  `wgs` workgroups (512 threads, a multiple of 8) each stream floats_per_wg floats out of one
 * of `regions` regions of region_floats floats; `share` workgroups of one XCD read the same region.  avg_us = average launch
 * duration over `iters` back-to-back launches. */
int cic_debug_stream_probe(const float* src, int64_t region_floats, int regions, int wgs, int floats_per_wg, int share,
                           float* sink, int iters, double* avg_us, cic_stream_t s);
/* diagnostics: per-workgroup phase stamps of the register-streaming GEMM (NULL = off) */
int cic_debug_set_stamps(unsigned long long* buf);
int cic_debug_set_attn_stamps(unsigned long long* buf);
int cic_debug_attn_cell_fused(int on);          /* 0: attention and att2ctx + cell of a decode step as two launches again */
/* diagnostics: per-workgroup phase stamps of the per-timestep GEMMs (rega / LDS-staged walker) and of the row kernels
 * (attention, sampler); NULL = off */
/* diagnostics, A/B timing of the GEMM dispatch: bit 0 clear = K-sliced tail tiles and K split over workgroups off (fixed
 * summation order); bits 8..15: 1 / 2 force 128x128 / 64x64 tiles; set bits turn a kernel family OFF: 16 strip walkers,
 * 21 16-wide walkers, 22 LDS-staged logit walker, 23 row-block split of 129..256-row products, 24 two-strip dX kernel;
 * bits 25..26: K parts per row tile of the logit walker (0 = two, 1 = its 4-wave form with one, 2 = four);
 * bit 27: the logit walker without its fused vocabulary epilogue (the row partials then come from cic_logit_partials);
 * bit 28: the LDS-tiled products on the f32-input MFMA only (no bf16-part kernel) */
int cic_debug_gemm_tail_split(int on);
/* diagnostics: cumulative phase times of gemm_bfx_kernel's K loop, [workgroup][wave][8] 100 MHz ticks: waiting at the first barrier,
 * store phase (incl. the wait for the tile's global loads), second barrier, load issue, MFMAs, K tiles, start, end; NULL = off */
int cic_debug_set_bfx_stamps(unsigned long long* buf);
/* diagnostics: 0 = the decode engines launch the attention query product on its own (A/B timing of the column split) */
int cic_debug_gates_att_fused(int on);
/* diagnostics: 0 = the decode engines run every step in full even after every caption has ended (A/B of the early stop) */
int cic_debug_early_stop(int on);
/* diagnostics: 0 = the teacher-forced recurrence (AttModel.forward) as three launches per step instead of spk_teacher_seq_kernel */
int cic_debug_teacher_seq(int on);
/* ... and 0 = the products of every BPTT step run although steps at or beyond the decode's length carry no gradient */
int cic_debug_bptt_early_stop(int on);
/* diagnostics: 0 = the speaker's BPTT loop as four launches per step instead of spk_bptt_seq_kernel (A/B timing, parity) */
int cic_debug_bptt_seq(int on);
/* diagnostics: phase stamps of spk_bptt_seq_kernel, [workgroup][step][8] s_memrealtime values (100 MHz) of lane 0: 0 step start,
 * 1 cell backward stored, 2 first hand-off passed, 3 products done + d att_res stored, 4 second hand-off passed, 5 attention
 * backward done, 6 third hand-off passed, 7 dh summed; NULL switches them off */
int cic_debug_set_bptt_stamps(unsigned long long* buf);
/* diagnostics: 2 (default) the whole GRU pass in one launch where the grid fits the chip, 1 one fused launch per step,
 * 0 every listener GRU step as a GEMM launch + a cell launch */
int cic_debug_gru_fused(int on);
/* diagnostics: phase stamps of the one-launch GRU BPTT loop (gru_seq_bwd_kernel): [workgroup][step][8] s_memrealtime values
 * (100 MHz) written by lane 0 - 0 step start, 1 gate derivative stored, 2 drained + barrier, 3 hand-off counter reached,
 * 4 MFMA chain done, 5 cross-wave sum barrier; NULL switches them off */
int cic_debug_set_gru_stamps(unsigned long long* buf);
/* diagnostics: 0 runs the speaker's a2c product and cell as two launches instead of the fused kernel */
int cic_debug_a2c_cell_fused(int on);
/* fault injection for the hand-offs of the one-launch recurrences (cic.h: "status word"): the spin bound in ticks of the
 * 100 MHz s_memrealtime counter (0 = back to 1 s), and ONE workgroup (blockIdx.x = wg) of the loops named by loop_bits
 * (CIC_STATUS_GRU_FWD | ...) that never counts itself in, so that its partners time out (loop_bits 0 = off) */
int cic_debug_spin_ticks(unsigned long long ticks);
int cic_debug_handoff_fault(int loop_bits, int wg);
/* the stand-in for a collective's kernel: `wgs` workgroups (256 threads) that each hold lds_bytes of LDS - 64 KB and more keep a
 * one-launch recurrence's workgroup (100+ KB) off their CU - for `ticks` of the 100 MHz clock, on stream s */
int cic_debug_hold_cus(int wgs, unsigned long long ticks, int lds_bytes, cic_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* CIC_DEV_H */
