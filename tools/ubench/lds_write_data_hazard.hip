// Two questions about the DATA registers of memory writes on gfx950, raised by anomalies in the f16x2 kernel
// (pixel-nerf-yolo_amd/csrc/mlp_h2.hip):
//  (1) can a vector-memory load issued right behind an LDS write, into the registers that hold the LDS write's data, land
//      before the LDS has read that data?  Sequence: ds_write_b64 [addr], v[a:a+1]; global_load_dwordx2 v[a:a+1], [hot line];
//      wait; read [addr] back -- in wave 0, with waves 1..7 of the workgroup idle or hammering the LDS with ds_read_b128.
//  (2) buffer_store_dwordx4 followed one instruction later by a VALU write of its first data register, with the constant 0 or
//      an SGPR in soffset, alone or behind three more stores.
// RESULT on MI355X: (1) 0 of 3.3e8 words wrong, idle or congested: not the mechanism of the gather anomaly.  (2) constant
// soffset: 25 % of the first dwords carry the new value -- the documented ">64-bit store data" hazard, for which LLVM inserts
// a wait state; SGPR soffset: 0 wrong in both forms -- as LLVM assumes.  The training stash nevertheless lost first dwords
// with exactly the SGPR form inside the fused kernel and stopped doing so with the constant form (where the compiler places
// the wait state); whatever else contributes there is not reproduced by this program.
// Build: hipcc --offload-arch=gfx950 -O2 -o lds_write_data_hazard.bin lds_write_data_hazard.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int ITERS = 20000;

__global__ __launch_bounds__(512) void k_lds(const unsigned* __restrict__ hot, unsigned* __restrict__ bad, int congest) {
    __shared__ __attribute__((aligned(16))) unsigned lds[16 * 1024];   // 64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16 * 1024; i += 512) lds[i] = 0;
    __syncthreads();
    if (wave == 0) {
        unsigned wrong = 0;
        const unsigned addr = (unsigned)(size_t)(&lds[2 * lane]) & 0xffff;   // byte address inside LDS (first 512 B)
        const unsigned* p = hot + 2 * lane;
        for (int it = 0; it < ITERS; ++it) {
            u32x2 v = {0x10000000u + (unsigned)it, 0x20000000u + (unsigned)lane};
            u32x2 back;
            asm volatile(
                "ds_write_b64 %2, %0\n\t"
                "global_load_dwordx2 %0, %3, off\n\t"
                "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
                "ds_read_b64 %1, %2\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "+v"(v), "=v"(back)
                : "v"(addr), "v"(p)
                : "memory");
            if (back[0] != 0x10000000u + (unsigned)it || back[1] != 0x20000000u + (unsigned)lane) ++wrong;
        }
        atomicAdd(bad, wrong);
    } else if (congest) {
        // keep the LDS pipeline full of 128-bit reads
        const unsigned a0 = (unsigned)(size_t)(&lds[1024 + 4 * tid]) & 0xffff;
        u32x4 acc = {0, 0, 0, 0};
        for (int it = 0; it < ITERS * 4; ++it) {
            u32x4 r;
            asm volatile("ds_read_b128 %0, %1 offset:0\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a0 + (unsigned)((it & 3) * 8192)) : "memory");
            acc += r;
        }
        if (acc[0] == 0x12345678u) bad[1] = acc[1];
    }
}

__global__ __launch_bounds__(64) void k_store(unsigned* __restrict__ out, unsigned* __restrict__ bad, int use_sgpr_soffset) {
    const int lane = threadIdx.x;
    const unsigned long long base = (unsigned long long)(out + (size_t)blockIdx.x * 64 * 4 * 16 * 2);
    const u32x4 rsrc = {(unsigned)base, (unsigned)(base >> 32) & 0xffffu, 64u * 16u * 16u * 2u, 0x00020000u};   // raw buffer descriptor
    unsigned wrong = 0;
    for (int it = 0; it < 2000; ++it) {
        const unsigned slot = (unsigned)(it & 15);
        const unsigned d0 = 0xA0000000u + (unsigned)it, d1 = 0xB0000000u + (unsigned)lane;
        const unsigned voff = (unsigned)lane * 16u, soff = slot * 1024u;
        if (use_sgpr_soffset == 2) {   // four stores back to back (as an epilogue issues them), then the overwrite
            asm volatile(
                "v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"
                "v_mov_b32 v44, %0\n\tv_mov_b32 v45, %1\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\ts_nop 4\n\t"
                "buffer_store_dwordx4 v[44:47], %2, %3, %5 offen\n\t"
                "buffer_store_dwordx4 v[44:47], %2, %3, %5 offen offset:1024\n\t"
                "buffer_store_dwordx4 v[44:47], %2, %3, %5 offen offset:2048\n\t"
                "buffer_store_dwordx4 v[40:43], %2, %3, %4 offen\n\t"
                "v_mov_b32 v40, 0x7e57da7a"
                :
                : "v"(d0), "v"(d1), "v"(voff), "s"(rsrc), "s"(soff), "s"(16u * 1024u)
                : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
        } else if (use_sgpr_soffset) {
            asm volatile(
                "v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\ts_nop 4\n\t"
                "buffer_store_dwordx4 v[40:43], %2, %3, %4 offen\n\t"
                "v_mov_b32 v40, 0x7e57da7a"          // VALU write of the first data register, one instruction later
                :
                : "v"(d0), "v"(d1), "v"(voff), "s"(rsrc), "s"(soff)
                : "memory", "v40", "v41", "v42", "v43");
        } else {
            asm volatile(
                "v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\ts_nop 4\n\t"
                "buffer_store_dwordx4 v[40:43], %2, %3, 0 offen\n\t"
                "v_mov_b32 v40, 0x7e57da7a"
                :
                : "v"(d0), "v"(d1), "v"(voff + soff), "s"(rsrc)
                : "memory", "v40", "v41", "v42", "v43");
        }
        __builtin_amdgcn_s_waitcnt(0);
        __threadfence();
        const unsigned got = __builtin_nontemporal_load(out + (size_t)blockIdx.x * 64 * 4 * 16 * 2 + slot * 256 + lane * 4);
        if (got != 0xA0000000u + (unsigned)it) ++wrong;
    }
    atomicAdd(bad, wrong);
}

int main() {
    unsigned *hot, *bad, *out;
    hipMalloc(&hot, 4096);
    hipMalloc(&bad, 16);
    hipMalloc(&out, (size_t)257 * 64 * 4 * 16 * 4 * 2);
    std::vector<unsigned> h(1024, 0xDEADBEEFu);
    hipMemcpy(hot, h.data(), 4096, hipMemcpyHostToDevice);
    for (int congest = 0; congest < 2; ++congest) {
        hipMemset(bad, 0, 16);
        hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 0, 0, hot, bad, congest);
        hipDeviceSynchronize();
        unsigned b = 0;
        hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
        printf("ds_write_b64 then global_load into its data registers, LDS %s: %u of %llu words read back wrong\n",
               congest ? "congested by 7 waves of ds_read_b128" : "idle", b, (unsigned long long)256 * 64 * ITERS);
    }
    for (int sg = 0; sg < 3; ++sg) {
        hipMemset(bad, 0, 16);
        hipLaunchKernelGGL(k_store, dim3(256), dim3(64), 0, 0, out, bad, sg);
        hipDeviceSynchronize();
        unsigned b = 0;
        hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
        printf("buffer_store_dwordx4 (%s soffset) then v_mov into its first data register: %u of %llu first dwords wrong\n",
               sg == 2 ? "SGPR, behind three more stores" : (sg ? "SGPR" : "constant 0"), b, (unsigned long long)256 * 64 * 2000);
    }
    return 0;
}
