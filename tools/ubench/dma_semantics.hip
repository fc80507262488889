// Where does `buffer_load_dwordx4 ... offen offset:IMM lds` put its data, and which bytes does it read?  (Is the instruction's immediate
// offset added to the LDS address, the memory address, or both?  Does M0 address LDS beyond 64 KiB?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
template <int IMM>
__global__ void probe(const unsigned *src, unsigned *dump, unsigned m0_base, int soff)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 64) reinterpret_cast<unsigned *>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(src), 0, 1 << 20, 0x00027000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + m0_base), 16, (int)(threadIdx.x * 16), soff, IMM, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 64) dump[i] = reinterpret_cast<unsigned *>(smem)[i];
}
template <int IMM>
void run(unsigned m0_base, int soff)
{
    unsigned *src, *dump;
    hipMalloc(&src, 1 << 20); hipMalloc(&dump, 160 * 1024);
    std::vector<unsigned> h((1 << 20) / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i * 4;         // value = byte offset
    hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)probe<IMM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe<IMM>, dim3(1), dim3(64), 160 * 1024, 0, src, dump, m0_base, soff);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned> d(160 * 1024 / 4);
    hipMemcpy(d.data(), dump, 160 * 1024, hipMemcpyDeviceToHost);
    int first = -1, last = -1;
    for (int i = 0; i < (int)d.size(); ++i) if (d[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; }
    printf("imm %4d  m0 %6u  soffset %5d : %s LDS bytes [%d, %d) <- source bytes starting at %u (lane 1 dword 0 holds %u)\n", IMM, m0_base, soff,
           e == hipSuccess ? "ok" : hipGetErrorString(e), first * 4, (last + 1) * 4, first >= 0 ? d[first] : 0, first >= 0 ? d[first + 4] : 0);
    hipFree(src); hipFree(dump);
}
int main()
{
    run<0>(2048, 0); run<1024>(2048, 0); run<0>(2048, 4096); run<1024>(2048, 4096); run<3072>(0, 0);
    run<0>(70000 & ~15u, 0); run<0>(140000 & ~15u, 512); run<2048>(131072, 0);
    return 0;
}
