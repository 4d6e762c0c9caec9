// Does an out-of-range raw buffer load retire ahead of an older in-range one?  ("wave-specialised GEMM experiment")
//     buffer_load_dword vA, <in-range, cold line>      ; older
//     buffer_load_dword vB, <offset 0xFFFFFFF0 / in-range control>   ; younger
//     s_waitcnt vmcnt(1)                               ; "all but the youngest have landed" if loads retire in issue order
//     v_mov vOut, vA                                    ; first reader of the older load's destination
// vA is preset to a poison value; a lane that still sees the poison behind the counted wait read the register before
// the older load landed.  Every lane reads its own cold line of a 1 GiB buffer (value = index), many waves per CU.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/oob_probe.bin scripts/oob_probe.hip && ./scripts/oob_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool OOB, bool DEEP>
__global__ __launch_bounds__(256) void probe(const unsigned int *buf, unsigned int nwords, unsigned int *bad, unsigned int *checked, int iters)
{
    u32x4 rs;
    const uint64_t a = (uint64_t)buf;
    rs[0] = (unsigned int)a;
    rs[1] = (unsigned int)(a >> 32);          // stride 0
    rs[2] = nwords * 4u;                      // num_records (bytes)
    rs[3] = 0x00020000u;
    unsigned int nbad = 0, nchk = 0;
    unsigned int idx = (blockIdx.x * 256u + threadIdx.x) * 977u;
    for (int it = 0; it < iters; ++it) {
        idx = (idx * 1664525u + 1013904223u);
        const unsigned int w = (idx >> 4) % nwords & ~31u;          // a 128-B line of its own
        const unsigned int offA = w * 4u;
        const unsigned int offB = OOB ? 0xFFFFFFF0u : ((w ^ 0x40000u) % nwords) * 4u;
        unsigned int out, vb;
        if (DEEP) {
            // the shape of the GEMM producer: 8 in-range 16-B loads of cold lines, 2 younger (out-of-range) ones, a counted
            // wait that leaves 9 in flight, first read of the oldest destination
            u32x4 r0, r1, r2, r3, r4, r5, r6, r7, y0, y1;
            const unsigned int o1 = ((w + 0x100000u) % nwords) * 4u, o2 = ((w + 0x200000u) % nwords) * 4u, o3 = ((w + 0x300000u) % nwords) * 4u;
            asm volatile("v_mov_b32 %0, 0xdeadbeef\n\t"
                         "s_nop 4\n\t"
                         "buffer_load_dwordx4 %1, %11, %15, 0 offen\n\t"
                         "buffer_load_dwordx4 %2, %12, %15, 0 offen\n\t"
                         "buffer_load_dwordx4 %3, %13, %15, 0 offen\n\t"
                         "buffer_load_dwordx4 %4, %14, %15, 0 offen\n\t"
                         "buffer_load_dwordx4 %5, %11, %15, 0 offen offset:128\n\t"
                         "buffer_load_dwordx4 %6, %12, %15, 0 offen offset:128\n\t"
                         "buffer_load_dwordx4 %7, %13, %15, 0 offen offset:128\n\t"
                         "buffer_load_dwordx4 %8, %14, %15, 0 offen offset:128\n\t"
                         "buffer_load_dwordx4 %9, %16, %15, 0 offen\n\t"
                         "buffer_load_dwordx4 %10, %16, %15, 0 offen\n\t"
                         "s_waitcnt vmcnt(9)"
                         : "=&v"(out), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(y0), "=&v"(y1)
                         : "v"(offA), "v"(o1), "v"(o2), "v"(o3), "s"(rs), "v"(offB)
                         : "memory");
            asm volatile("v_mov_b32 %0, %1" : "=v"(out) : "v"(r0[0]));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(y0), "+v"(y1) :: "memory");
            vb = r1[0] ^ r2[0] ^ r3[0] ^ r4[0] ^ r5[0] ^ r6[0] ^ r7[0] ^ y0[0] ^ y1[0];
            if (r0[0] != w) ++nbad;           // r0 itself must be right after vmcnt(0)
        } else
        asm volatile("v_mov_b32 %0, 0xdeadbeef\n\t"
                     "s_nop 4\n\t"
                     "buffer_load_dword %0, %2, %4, 0 offen\n\t"
                     "buffer_load_dword %1, %3, %4, 0 offen\n\t"
                     "s_waitcnt vmcnt(1)\n\t"
                     "v_mov_b32 %0, %0\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(out), "=&v"(vb)
                     : "v"(offA), "v"(offB), "s"(rs)
                     : "memory");
        ++nchk;
        if (out != w) ++nbad;
        if (vb == 0x12345678u) ++nbad;        // keep vb alive
    }
    atomicAdd(bad, nbad);
    atomicAdd(checked, nchk);
}

int main()
{
    const unsigned int nwords = 256u << 20;   // 1 GiB
    unsigned int *buf, *cnt;
    hipMalloc(&buf, (size_t)nwords * 4);
    hipMalloc(&cnt, 16);
    std::vector<unsigned int> h(1 << 20);
    // value = word index (filled on the device side would be simpler; a host loop over 1 GiB is fine once)
    for (unsigned int base = 0; base < nwords; base += (1u << 20)) {
        for (unsigned int i = 0; i < (1u << 20); ++i) h[i] = base + i;
        hipMemcpy(buf + base, h.data(), (size_t)4 << 20, hipMemcpyHostToDevice);
    }
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(cnt, 0, 16);
        if (mode == 0) hipLaunchKernelGGL((probe<false, false>), dim3(256 * 8), dim3(256), 0, 0, buf, nwords, cnt, cnt + 1, 2000);
        else if (mode == 1) hipLaunchKernelGGL((probe<true, false>), dim3(256 * 8), dim3(256), 0, 0, buf, nwords, cnt, cnt + 1, 2000);
        else if (mode == 2) hipLaunchKernelGGL((probe<false, true>), dim3(256 * 8), dim3(256), 0, 0, buf, nwords, cnt, cnt + 1, 500);
        else hipLaunchKernelGGL((probe<true, true>), dim3(256 * 8), dim3(256), 0, 0, buf, nwords, cnt, cnt + 1, 500);
        hipDeviceSynchronize();
        unsigned int r[2];
        hipMemcpy(r, cnt, 8, hipMemcpyDeviceToHost);
        printf("%s, %s younger load(s): %u of %u reads behind the counted wait saw the register before the older load landed\n",
               mode & 2 ? "8 x 16 B + 2" : "1 x 4 B + 1 ", mode & 1 ? "out-of-range" : "in-range    ", r[0], r[1]);
    }
    return 0;
}
