// micro-check: does `buffer_load_dwordx4 ... lds` write ZEROS to LDS for lanes whose offset is out of range?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, float* out, int n)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
    for (int i = threadIdx.x; i < 1024; i += 256) ((float*)lds)[i] = -7.0f;     // poison
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
    int voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = (int)0x80000000;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (threadIdx.x >> 6) * 1024), 16,
                                             voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = ((float*)lds)[i];
}
int main()
{
    float *in, *out, h[1024], hin[1024];
    for (int i = 0; i < 1024; ++i) hin[i] = (float)(i + 1);
    hipMalloc(&in, 4096); hipMalloc(&out, 4096);
    hipMemcpy(in, hin, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, in, out, 1024);
    hipMemcpy(h, out, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t)
        for (int j = 0; j < 4; ++j) {
            const float want = (t & 1) ? 0.0f : (float)(t * 4 + j + 1);
            if (h[t * 4 + j] != want) { if (bad < 8) printf("lane %d elt %d got %g want %g\n", t, j, h[t * 4 + j], want); ++bad; }
        }
    printf("glds_oob: %s (%d mismatches)\n", bad ? "FAIL" : "OK zeros for OOB lanes", bad);
    return 0;
}
