/* tools/csrc/skp_lab.h -- C ABI of libskp_lab.so: measurement aids kept OUT of the product library
 * (libskp_hip.so, include/skp.h).  Same conventions: extern "C", caller-owned device buffers, caller's stream, int return
 * (0, a positive hipError_t, or a negative SKP_E_* argument error). */
#ifndef SKP_LAB_H
#define SKP_LAB_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Measurement aid (bench.py): sustained rate, in TFLOP/s, of back-to-back independent v_mfma_f32_16x16x4_f32 with
 * `waves_per_simd` (1..4) resident waves on every SIMD.  Synchronises `stream`.  scratch: >= 256*4*256 floats. */
int skp_probe_mfma_f32(int waves_per_simd, int iters, float* scratch, float* tflops, void* stream);

#ifdef __cplusplus
}
#endif
#endif
