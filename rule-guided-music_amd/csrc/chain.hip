// chain.hip -- the 28 blocks of a DiTRotary forward as ONE persistent launch (round 5; guided_diffusion/dit.py:332-336, 618-634).
//
// Why: at the batch BASELINE's metric is quoted on (B = 16, M = 4096 rows) every GEMM of a block is ONE round of 256x256 tiles -- all 224-256
// workgroups of a launch meet in its cold-instruction-cache prologue and again in its write-bound epilogue, fc1's 288 tiles leave a 32-tile
// tail launch, fc2's K slices a reduce launch, and seven launch boundaries per block drain the chip (the dominant kernel at
// 0.47 of peak, the whole forward at 0.36).  Nothing between launches can fix a forward whose every GEMM is exactly one round.
//
// What: 256 resident workgroups (one per CU: 256 threads, 512 registers, 160 KiB of LDS -- the shape of the one-wave-per-SIMD GEMM) walk ONE
// static list of work items for the whole forward, claimed in list order from a device counter:
//     qkv tile -> attention (sample, head) -> proj tile -> LayerNorm rows -> fc1 tile -> fc2 K-slice tile -> reduce + next LayerNorm rows
// The rows of a sample (T = 256 = the GEMM tile height) never meet another sample's inside a block, so readiness is SAMPLE-granular: one
// monotonic counter per sample counts its finished items; an item waits (one lane, relaxed agent-scope poll + s_sleep, bounded) until the
// counter reaches the number of items of all the sample's earlier phases.  Every dependency of an item precedes it in the list, so a
// waiting workgroup only ever waits for items that running workgroups hold: no deadlock whatever the residency.
//
// Hand-off (MI355X_MICROARCH.md "publish-large", common.h store16_sc1): every output of an item is stored device-coherent (write-through),
// the workgroup waits vmcnt(0), meets at a barrier and adds 1 to the sample's counter; a consumer polls, executes ONE agent-scope acquire
// fence (L1 invalidate), meets at a barrier and reads with plain loads / LDS-DMA.  No cache write-back anywhere.
//
// Arithmetic: the item bodies ARE the one-launch kernels' bodies (gemm2_body.h, attention_x3_body.h, ln_body.h) -- a chained forward
// differs from the launch-per-GEMM forward only where the latter picks another algorithm: its key-blocked attention (running-maximum
// softmax; here the single-pass kernel) and proj / fc1's tail on 128x64 tiles (same per-element sums: bit-identical).
#include <stdint.h>
#include <vector>
#include "common.h"
#include "gemm2_body.h"
#include "attention_x3_body.h"
#include "ln_body.h"

namespace rgm {

constexpr int CHAIN_LDS = 160 * 1024;             // all of a CU's LDS: one workgroup per CU
constexpr int CHAIN_CTRL = CHAIN_LDS - 64;        // the claimed item, behind the largest image an item body keeps (attention: 160 896 B)
constexpr unsigned long long CHAIN_WAIT_TICKS = 50000000ull;   // 0.5 s of the 100 MHz clock: a dependency that never arrives ends the launch

// Row items (LayerNorm / K-slice reduce + LayerNorm): the item's rows are dealt to the four waves, row0 + wave + 4 j.  A wave requests
// several rows before it finishes the first -- one row at a time a 16-row item took 21 us (LayerNorm) / 31 us (reduce): four round trips
// of a lone wave in series (tools/chain_spans.py, profiles/r05_chain_spans_v1.txt).  All rows of an item belong to one sample.
template <int MAXV>
__device__ __forceinline__ void chain_rows(const GemmParams& g, const float* __restrict__ P, int rows_per_item, int kind, int row0, int wave, int lane) {
  LnMod<MAXV> md;                  // the sample's modulation row (shift, scale): once per item
  if (kind != CHAIN_REDUCE) ln_mod_load_mod<MAXV>(md, g.ln_shift, g.ln_scale, g.N, (long long)(row0 / g.ln_rows_per_batch) * g.ln_mod_ld, lane);
  if (kind == CHAIN_LN) {
    constexpr int R = 4;
#pragma unroll 1
    for (int r = wave; r < rows_per_item; r += 4 * R) {
      LnRow<MAXV> st[R];
#pragma unroll
      for (int j = 0; j < R; ++j) ln_mod_load<MAXV>(st[j], g.A, min(row0 + r + 4 * j, g.M - 1), g.N, lane);
#pragma unroll
      for (int j = 0; j < R; ++j)
        if (r + 4 * j < rows_per_item && row0 + r + 4 * j < g.M)
          ln_mod_finish<MAXV, 1, 1>(st[j], g.ln_out, row0 + r + 4 * j, g.N, g.ln_eps, nullptr, nullptr, g.ln_shift, g.ln_scale, g.ln_mod_ld, g.ln_rows_per_batch, 1, md, lane);
    }
  } else {
    RedShared<MAXV> sh;
    splitk_reduce_shared<MAXV>(sh, g, row0, lane);
#pragma unroll 1
    for (int r = wave; r < rows_per_item; r += 4) {
      const int row = row0 + r;
      if (row >= g.M) break;
      RedRow<MAXV, 3> st;
      splitk_reduce_load<MAXV, 3>(st, P, g, row, lane);
      if (kind == CHAIN_REDUCE_LN) splitk_reduce_ln_finish<MAXV, 3, 1, 1, 1>(st, sh, g, row, md, lane);
      else splitk_reduce_ln_finish<MAXV, 3, 1, 0>(st, sh, g, row, LnMod<MAXV>{}, lane);
    }
  }
}

// a pointer read from a descriptor in memory is a generic pointer to the compiler (flat loads, lgkmcnt + vmcnt waits); through address
// space 1 and back the address-space inference sees global memory, as it does for kernel arguments
template <class Tp>
__device__ __forceinline__ Tp* as_global(Tp* q) {
  return (Tp*)(__attribute__((address_space(1))) Tp*)q;
}
// the descriptor's GemmParams as a LOCAL copy (uniform scalar loads, once per item; unused fields fall away) with global pointers
__device__ __forceinline__ GemmParams chain_params(const GemmParams& m) {
  GemmParams g = m;
  g.A = as_global(g.A); g.B = as_global(g.B); g.C = as_global(g.C); g.bias = as_global(g.bias); g.gate = as_global(g.gate);
  g.res = as_global(g.res); g.ln_out = as_global(g.ln_out); g.ln_shift = as_global(g.ln_shift); g.ln_scale = as_global(g.ln_scale);
  g.aux = nullptr; g.C2 = nullptr; g.stats = nullptr; g.gn_count = nullptr; g.gn_fail = nullptr; g.gn_gamma = nullptr; g.gn_beta = nullptr;
  g.ln_done = nullptr; g.sk_ws = nullptr; g.aload = 0; g.conv_kmajor = 0;
  return g;
}

__global__ __launch_bounds__(256) void dit_chain_kernel(const ChainOp* __restrict__ ops, const uint4* __restrict__ items, const int n_items,
                                                        unsigned* __restrict__ ctl, const char* __restrict__ zero_page, const int trace,
                                                        unsigned long long* __restrict__ times) {
  extern __shared__ __attribute__((aligned(16))) char ring[];
  volatile int* ctrl = reinterpret_cast<volatile int*>(ring + CHAIN_CTRL);
  unsigned* progress = ctl + CHAIN_CTL_PROGRESS;
  // trace (rgm_dit_chain_peek, debugging): where every workgroup is -- { item index + 1, 1 waiting / 2 running / 3 publishing / 4 left }
  unsigned* tr = ctl + CHAIN_CTL_TRACE + 2 * blockIdx.x;
  // ONE lane-0 region per iteration, in front of a barrier, and ONE back edge: with a second `if (tid == 0)` at the loop's tail the
  // compiler split the loop in two (one per back edge) and parked lane 0 -- masked, waiting for the other lanes to leave the inner loop --
  // in front of the counter update for ever, while the rest of the workgroup re-ran its item (the first run of this kernel hung that way).
  int prev_grp = -1;                                 // the item this workgroup finished last: published at the top of the next iteration
  unsigned prev_idx = 0;
  for (;;) {
    // the thread index behind an opaque copy, once per item: everything an item body derives from it is then computed inside the iteration.
    // Derived from threadIdx.x directly, the loop-invariant part of every body's index arithmetic was hoisted in front of the loop, had to
    // live through the 512-register GEMM body in scratch, and every reload (a VMEM load: s_waitcnt vmcnt(0)) sat between the row items'
    // global loads -- five serial round trips where one was written (a 16-row LayerNorm item: 15 us)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    if (tid == 0) {
      if (prev_grp >= 0) __hip_atomic_fetch_add(progress + prev_grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // times (rgm_dit_chain_times, tools/chain_spans.py): per item { claimed, dependencies met, finished, workgroup } on the 100 MHz clock
      if (times && prev_grp >= 0) times[8 * (size_t)prev_idx + 2] = __builtin_amdgcn_s_memrealtime();
      const unsigned long long t_claim = times ? __builtin_amdgcn_s_memrealtime() : 0ull;
      const unsigned idx = __hip_atomic_fetch_add(&ctl[CHAIN_CTL_HEAD], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      prev_idx = idx;
      int w0 = -1, w1 = 0;
      if (trace) {
        __hip_atomic_store(tr, idx + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(tr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (idx < (unsigned)n_items && __hip_atomic_load(&ctl[CHAIN_CTL_ERROR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        const uint4 it = items[idx];
        w0 = (int)it.x;
        w1 = (int)it.y;
        const unsigned need = it.z;
        if (need) {
          unsigned* cnt = progress + (it.x >> 16);
          const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
          while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > CHAIN_WAIT_TICKS) {       // never on a healthy run: report the item and stop claiming
              __hip_atomic_store(&ctl[CHAIN_CTL_ERROR], idx + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              w0 = -1;
              break;
            }
            if (__hip_atomic_load(&ctl[CHAIN_CTL_ERROR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {   // somebody else gave up
              w0 = -1;
              break;
            }
            __builtin_amdgcn_s_sleep(8);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                      // this CU's L1 holds nothing older than the counter value
        }
      }
      ctrl[0] = w0;
      ctrl[1] = w1;
      if (times && w0 >= 0) {
        times[8 * (size_t)idx] = t_claim;
        times[8 * (size_t)idx + 1] = __builtin_amdgcn_s_memrealtime();
        times[8 * (size_t)idx + 3] = blockIdx.x;
      }
      if (trace) __hip_atomic_store(tr + 1, w0 < 0 ? 4u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int w0 = __builtin_amdgcn_readfirstlane(ctrl[0]);
    const int w1 = __builtin_amdgcn_readfirstlane(ctrl[1]);
    if (w0 < 0) break;
    const int opi = w0 & 0xffff, grp = (w0 >> 16) & 0xffff, sub = w1 & 0xffff, z = (w1 >> 16) & 0xffff;
    const ChainOp& op = ops[opi];
    const int kind = __builtin_amdgcn_readfirstlane(op.kind);
    const GemmParams gp = chain_params(op.g);
    if (kind == CHAIN_GEMM) {
      const int tn = op.tiles_n;
      gemm2_body<256, 256, 2, 2, 0, 2, 0, 5, 1>(gp, zero_page, 0, tn, 0, nullptr, grp * tn + sub, z, -1, tid);
    } else if (kind == CHAIN_ATTN) {
      const int heads = op.heads;
      attn_x3_body<72, 8, 1>(ring, gp.A, gp.C, as_global(op.cos_tab), as_global(op.sin_tab), op.T, heads, op.rot_half, nullptr, 1, 1, grp * heads + sub, tid);
    } else {
      const int rpi = op.rows_per_item;
      chain_rows<5>(gp, as_global(op.P), rpi, kind, grp * op.rows_per_group + sub * rpi, tid >> 6, tid & 63);
    }
    // publish: the item's stores are device-coherent and complete when vmcnt retires them; only then may the sample's counter move
    // (lane 0 adds to it first thing behind this barrier, at the top of the next iteration)
    int tid_b = threadIdx.x;                           // (a second opaque copy: nothing of the first lives through the body)
    asm volatile("" : "+v"(tid_b));
    if (times && tid_b == 0) times[8 * (size_t)prev_idx + 4] = __builtin_amdgcn_s_memrealtime();     // wave 0 left the body
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (times && tid_b == 0) times[8 * (size_t)prev_idx + 5] = __builtin_amdgcn_s_memrealtime();     // ... and its stores have retired
    __syncthreads();
    prev_grp = grp;
  }
}

static char* g_chain_zero_page = nullptr;
static long long g_chain_launches = 0;

int dit_chain_launch(const ChainOp* d_ops, const uint4* d_items, int n_items, unsigned* d_ctl, int n_groups, hipStream_t s,
                     unsigned long long* d_times) {
  RGM_REQUIRE(d_ops && d_items && d_ctl && n_items > 0 && n_groups > 0 && n_groups <= CHAIN_MAX_GROUPS, "dit_chain: bad arguments");
  if (!g_chain_zero_page) {
    RGM_CHECK_HIP(hipMalloc(&g_chain_zero_page, 4096));
    RGM_CHECK_HIP(hipMemset(g_chain_zero_page, 0, 4096));
  }
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    RGM_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    RGM_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    RGM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dit_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CHAIN_LDS));
    cus = prop.multiProcessorCount;
  }
  static const int trace = getenv("RGM_CHAIN_TRACE") ? atoi(getenv("RGM_CHAIN_TRACE")) : 0;
  RGM_REQUIRE(cus <= CHAIN_MAX_CUS, "dit_chain: %d CUs", cus);
  RGM_CHECK_HIP(hipMemsetAsync(d_ctl, 0, sizeof(unsigned) * (trace ? CHAIN_CTL_WORDS : CHAIN_CTL_PROGRESS + n_groups), s));
  hipLaunchKernelGGL(dit_chain_kernel, dim3(cus), dim3(256), CHAIN_LDS, s, d_ops, d_items, n_items, d_ctl, (const char*)g_chain_zero_page, trace, d_times);
  RGM_LAUNCH_CHECK();
  ++g_chain_launches;
  return RGM_OK;
}

long long dit_chain_launch_count() { return g_chain_launches; }

}  // namespace rgm
