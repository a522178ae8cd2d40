#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const char* x, int n, int soff, float* out) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n, 0x00020000);
    int voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0x7fffffff;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)smem, 16, voff, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + 1024), 16, voff, soff, 0, 0);
    __syncthreads();
    out[threadIdx.x] = ((float*)smem)[threadIdx.x * 4] + ((float*)smem)[256 + threadIdx.x * 4];
}
int main() {
    char* x; float* o; 
    hipMalloc(&x, 1 << 20); hipMalloc(&o, 4096);
    float* h = (float*)malloc(1 << 20);
    for (int i = 0; i < (1 << 18); ++i) h[i] = (float)i;
    hipMemcpy(x, h, 1 << 20, hipMemcpyHostToDevice);
    k<<<1, 64, 4096>>>(x, 512, 8192, o);     // num_records 512 bytes: lanes >= 32 out of range on first load
    float r[64]; hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("%g ", r[i]); printf("\n");
    return 0;
}
