// Which stream pairs let a small kernel run UNDER a wide one?  (scripts/lane_probe.py; DESIGN.md section 4, "Two frames in flight")
// wide:  many single-wave workgroups held to a few per CU by their LDS, each spinning a little -- the kernel's dispatch stays
//        open for its whole duration while most of the chip's wave slots stay free;
// small: a few hundred workgroups spinning longer.
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/ubench/lane_probe.hip -o /tmp/liblaneprobe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void k_spin(int spins, int lds_words, unsigned *sink) {
    extern __shared__ unsigned s_pad[];
    if (lds_words > 0 && threadIdx.x == 0) s_pad[0] = (unsigned)spins;
    unsigned acc = 0;
    for (int i = 0; i < spins; ++i) {
        __builtin_amdgcn_s_sleep(32);
        acc += (unsigned)i;
    }
    if (acc == 0xffffffffu) *sink = acc;   // (never: keeps the loop)
}

extern "C" int lane_probe_launch(void *stream, int blocks, int spins, int lds_bytes, void *sink) {
    if (lds_bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_spin), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(k_spin, dim3((unsigned)blocks), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, spins, lds_bytes / 4, (unsigned *)sink);
    return (int)hipGetLastError();
}
