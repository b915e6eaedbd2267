// Probe: what does `buffer_load_dwordx4 ... lds` (LDS-DMA) write for lanes whose offset is out of the buffer's range?
// The implicit-GEMM staging of conv_glds_f16_kernel relies on zeros (padding taps), as the register-staged kernels rely
// on buffer_load returning zeros.  hipcc --offload-arch=gfx950 tools/probes/glds_oob.hip -o /tmp/glds_oob && /tmp/glds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const float *x, float *y, unsigned nbytes, int oob)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, nbytes, 0x00020000);
    for (int i = threadIdx.x; i < 1024; i += 64) ((float *)lds)[i] = -7.f;
    __syncthreads();
    unsigned off = threadIdx.x * 16;
    if (oob && (threadIdx.x & 1)) off = nbytes;     // odd lanes out of range
    if (oob == 2 && (threadIdx.x & 1)) off = 0xfffffff0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) y[i] = ((float *)lds)[i];
}

int main()
{
    float *x, *y;
    std::vector<float> hx(256), hy(256);
    for (int i = 0; i < 256; ++i) hx[i] = (float)(i + 1);
    hipMalloc(&x, 1024); hipMalloc(&y, 1024);
    hipMemcpy(x, hx.data(), 1024, hipMemcpyHostToDevice);
    for (int oob = 0; oob < 3; ++oob) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, 1024u, oob);
        hipMemcpy(hy.data(), y, 1024, hipMemcpyDeviceToHost);
        int same = 0, zero = 0, kept = 0, other = 0;
        for (int i = 0; i < 256; ++i) {
            const bool odd = (i / 4) & 1;
            if (!oob || !odd) { same += hy[i] == hx[i]; continue; }
            if (hy[i] == 0.f) ++zero; else if (hy[i] == -7.f) ++kept; else ++other;
        }
        std::printf("oob=%d: in-range floats correct %d; out-of-range floats: zero %d, untouched %d, other %d\n", oob, same, zero, kept, other);
    }
    return 0;
}
