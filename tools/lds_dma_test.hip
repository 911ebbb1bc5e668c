// Does a buffer load to LDS write 0.0f for lanes whose offset is out of range?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const float* x, float* y, int n) {
    __shared__ float buf[256];
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, n * 4, 0x00020000);
    int off = (threadIdx.x % 5 != 0) ? threadIdx.x * 4 : (int)0x80000000;
    buf[threadIdx.x] = -1.0f; buf[threadIdx.x + 64] = -1.0f;
    __syncthreads();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)buf, 4, off, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(buf + 64), 4, off, 128, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    y[threadIdx.x] = buf[threadIdx.x];
    y[threadIdx.x + 64] = buf[threadIdx.x + 64];
}
int main() {
    float h[256], *x, *y;
    for (int i = 0; i < 256; ++i) h[i] = 100 + i;
    (void)hipMalloc(&x, 1024); (void)hipMalloc(&y, 1024);
    (void)hipMemcpy(x, h, 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(x, y, 96);                        // 96 records: second load (soffset 128 B) runs off the end for lanes >= 64
    (void)hipMemcpy(h, y, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 128; ++i) printf("%g%c", h[i], i % 16 == 15 ? '\n' : ' ');
    return 0;
}
