// Short-sequence attention co-residency hazard: which of {placement in the upper half of a CU's LDS, stale LDS
// contents, a sibling workgroup of the same launch, partially retired workgroups} makes rotary_attention_x3_kernel<72,4> return
// wrong rows at T = 128?  The probe launches the library's own kernel (this file includes attention_x3.hip) with an explicit block
// size and LDS request, records per workgroup where it ran (HW_ID, LDS_ALLOC, XCC_ID), when, and what its staging barrier saw.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRGM_ATTN_HAZARD_DBG -I../../rule-guided-music_amd/csrc -o attn_hazard attn_hazard.hip
//        ... (no flag: the product kernel since round 5) -o attn_hazard_two_phase   (the Q prologue with every load retired before the first use)
//        ... -DRGM_ATTN_HAZARD_DUMP -o attn_hazard_dump             (+ the Q fragments of every lane; tools/ubench/attn_hazard.sh builds all three)
// run:   ./attn_hazard [launches per experiment, default 40] [N, default 48] [substring of the experiment names to run]
// Whether the failure shows depends on the exact instruction schedule of the kernel: a build reproduces it in about half of its launches
// (N = 96) or never.  The narrowing recorded in DESIGN 4h was made with the builds of commit a6d844e (profiles/r04_attn_hazard_*.txt).
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "attention_x3.hip"

// ---- what attention_x3.hip expects from the rest of the library
namespace rgm {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
int rotary_attention_launch(const float*, float*, const float*, const float*, int, int, int, int, int, hipStream_t, float*, int) { return -1; }
int attn_split_mode() { return 0; }
}  // namespace rgm
extern "C" int rgm_get_gemm_precision(void) { return 1; }

#define CK(x)                                                                                 \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);             \
      exit(2);                                                                                \
    }                                                                                         \
  } while (0)

// every dword of a CU's 160 KiB LDS <- a NaN pattern (as fp32 and as two bf16); one workgroup per CU by construction (needs all of it)
__global__ __launch_bounds__(1024) void lds_fill_kernel(unsigned pattern, unsigned* where) {
  extern __shared__ unsigned lds_all[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) lds_all[i] = pattern;
  __syncthreads();
  if (threadIdx.x == 0) {
    where[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    where[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 4000) __builtin_amdgcn_s_sleep(32);   // 40 us: all 256 are resident together
  }
  __syncthreads();
  if (lds_all[threadIdx.x] != pattern) where[0] = 0xdeadbeef;
}

// holds `lds` bytes of a CU's LDS (the LOWER part when it is first on the CU) until *flag != 0 or ~20 ms
__global__ __launch_bounds__(64) void lds_holder_kernel(const int* flag, unsigned* where) {
  extern __shared__ unsigned lds_all[];
  lds_all[threadIdx.x] = threadIdx.x;
  if (threadIdx.x == 0) {
    where[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    where[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 6);
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && __builtin_amdgcn_s_memrealtime() - t0 < 2000000)
    __builtin_amdgcn_s_sleep(64);
  if (lds_all[threadIdx.x] != threadIdx.x) where[0] = 0xdeadbeef;
}

static float frand(unsigned long long& s) {   // approx N(0,1): sum of 12 uniforms - 6
  float a = 0.f;
  for (int i = 0; i < 12; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    a += (float)((s >> 40) & 0xffffff) / 16777216.0f;
  }
  return a - 6.0f;
}

constexpr int HD = 72, NKT = 4, HEADS = 16, T = 128, D = HEADS * HD, ROT_HALF = 18;
static int N = 48;

struct Setup {
  float *qkv, *out, *cosd, *sind;
  rgm::AttnDbg* dbg;
  int* cnt;
  unsigned* where;
  int* flag_host;   // pinned, device-visible
  std::vector<float> ref, cur, dref, dcur, hq, hc, hs;
  float* dump = nullptr;
  std::vector<rgm::AttnDbg> hdbg;
};

static unsigned cu_key(const rgm::AttnDbg& d) {   // (xcc, se, sh, cu)
  const unsigned cu = (d.hw_id >> 8) & 0xf, sh = (d.hw_id >> 12) & 1, se = (d.hw_id >> 13) & 7;
  return ((d.xcc_id & 0xf) << 12) | (se << 8) | (sh << 4) | cu;
}

static void launch_attn(Setup& S, int threads, size_t lds, hipStream_t st, int out_split = 0) {
  auto kern = rgm::rotary_attention_x3_kernel<HD, NKT>;
  hipLaunchKernelGGL(kern, dim3(N * HEADS), dim3(threads), lds, st, S.qkv, S.out, S.cosd, S.sind, T, HEADS, ROT_HALF, (float*)nullptr, out_split, 1);
  CK(hipGetLastError());
}

struct Opts {
  const char* name;
  int threads;
  size_t lds;
  bool nanfill = false, count = false, holders = false;
  size_t holder_lds = 81024;
  int mode = 0;   // g_attn_mode of the debug build
  int out_split = 0;
  bool dump = false;
};

static void experiment(Setup& S, const Opts& o, int launches) {
  const int grid = N * HEADS;
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  int wrong_launches = 0, wrong_blocks = 0, nan_blocks = 0, short_arrivals = 0, upper_blocks = 0, total_blocks = 0, wrong_upper = 0, wrong_lower = 0;
  std::map<unsigned, int> alloc_hist, alloc_wrong;
  int printed = 0;
  int* cntp = o.count ? S.cnt : nullptr;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(rgm::g_attn_cnt), &cntp, sizeof(cntp)));
  const size_t dump_n = (size_t)grid * 4 * ATTN_DUMP_ITEMS * 64;
  float* dumpp = o.dump ? S.dump : nullptr;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(rgm::g_attn_dump), &dumpp, sizeof(dumpp)));
  {   // this experiment's reference: the same kernel, one workgroup per CU, plain exchanges
    launch_attn(S, o.dump ? o.threads : 512, 80 * 1024 + 512, sa, o.out_split);
    CK(hipStreamSynchronize(sa));
    CK(hipMemcpy(S.ref.data(), S.out, (size_t)N * T * D * 4, hipMemcpyDeviceToHost));
    if (o.dump) {
      S.dref.resize(dump_n);
      S.dcur.resize(dump_n);
      CK(hipMemcpy(S.dref.data(), S.dump, dump_n * 4, hipMemcpyDeviceToHost));
    }
  }
  for (int l = 0; l < launches; ++l) {
    CK(hipMemsetAsync(S.out, 0xff, (size_t)N * T * D * 4, sa));
    CK(hipMemsetAsync(S.dbg, 0, sizeof(rgm::AttnDbg) * grid, sa));
    if (o.count) CK(hipMemsetAsync(S.cnt, 0, sizeof(int) * grid, sa));
    if (o.nanfill) {
      hipLaunchKernelGGL(lds_fill_kernel, dim3(256), dim3(1024), 160 * 1024, sa, 0x7fc07fc0u, S.where);
      CK(hipGetLastError());
    }
    CK(hipStreamSynchronize(sa));
    if (o.nanfill && l == 0) {
      std::vector<unsigned> w(512);
      CK(hipMemcpy(w.data(), S.where, 512 * 4, hipMemcpyDeviceToHost));
      std::map<unsigned, int> cus;
      for (int i = 0; i < 256; ++i) {
        rgm::AttnDbg d{};
        d.hw_id = w[2 * i];
        d.xcc_id = w[2 * i + 1];
        cus[cu_key(d)]++;
      }
      printf("  [%s] LDS fill covered %zu distinct CUs with 256 workgroups\n", o.name, cus.size());
    }
    if (o.holders) {
      *S.flag_host = 0;
      hipLaunchKernelGGL(lds_holder_kernel, dim3(256), dim3(64), o.holder_lds, sb, S.flag_host, S.where);
      CK(hipGetLastError());
      // give the holders time to become resident on every CU before the attention launch is queued
      usleep(3000);
    }
    launch_attn(S, o.threads, o.lds, sa, o.out_split);
    CK(hipStreamSynchronize(sa));
    if (o.holders) {
      *S.flag_host = 1;
      CK(hipStreamSynchronize(sb));
      if (l == 0) {
        std::vector<unsigned> w(512);
        CK(hipMemcpy(w.data(), S.where, 512 * 4, hipMemcpyDeviceToHost));
        std::map<unsigned, int> cus, allocs;
        for (int i = 0; i < 256; ++i) {
          rgm::AttnDbg d{};
          d.hw_id = w[2 * i];
          cus[(d.hw_id >> 8) & 0xff]++;
          allocs[w[2 * i + 1]]++;
        }
        printf("  [%s] holders: LDS_ALLOC values:", o.name);
        for (auto& kv : allocs) printf(" %08x:%d", kv.first, kv.second);
        printf("\n");
      }
    }
    CK(hipMemcpy(S.cur.data(), S.out, (size_t)N * T * D * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(S.hdbg.data(), S.dbg, sizeof(rgm::AttnDbg) * grid, hipMemcpyDeviceToHost));
    // first workgroup on each CU by start time: decides what "lower" LDS_ALLOC looks like
    bool any = false;
    for (int b = 0; b < grid; ++b) {
      const int n = b / HEADS, h = b % HEADS;
      const rgm::AttnDbg& d = S.hdbg[b];
      const unsigned base = d.lds_alloc & 0xfff;   // printed raw as well
      alloc_hist[d.lds_alloc]++;
      ++total_blocks;
      if (base != 0) ++upper_blocks;
      if (o.count && d.arrivals != (unsigned)(o.threads / 64)) ++short_arrivals;
      int nw = 0, nn = 0, nw_h[2] = {0, 0};
      float mx = 0.f;
      unsigned tiles = 0;
      double rmin = 1e30, rmax = -1e30, row_spread = 0;
      unsigned long long rowmask[2] = {0, 0};
      for (int r = 0; r < T; ++r) {
        const size_t off = ((size_t)n * T + r) * D + h * HD;
        double qmin = 1e30, qmax = -1e30;
        for (int c = 0; c < HD; ++c) {
          const float a = S.cur[off + c], e = S.ref[off + c];
          if (memcmp(&a, &e, 4) != 0) {
            ++nw;
            rowmask[r >> 6] |= 1ull << (r & 63);
            ++nw_h[(c >> 2) & 1];                 // register layout of O^T: channel d belongs to lanes of half (d / 4) % 2
            tiles |= 1u << (r >> 5);
            if (a != a) ++nn;
            else {
              mx = fmaxf(mx, fabsf(a - e));
              if (fabsf(e) > 1e-3f) {
                const double ratio = (double)a / e;
                qmin = fmin(qmin, ratio);
                qmax = fmax(qmax, ratio);
              }
            }
          }
        }
        if (qmax > -1e29) {
          rmin = fmin(rmin, qmin);
          rmax = fmax(rmax, qmax);
          row_spread = fmax(row_spread, qmax - qmin);   // ~0: every wrong channel of a query is off by ONE factor (a wrong 1/sum)
        }
      }
      if (nw) {
        any = true;
        ++wrong_blocks;
        alloc_wrong[d.lds_alloc]++;
        if (base != 0) ++wrong_upper; else ++wrong_lower;
        if (nn) ++nan_blocks;
        // sibling(s): same CU, overlapping in time
        std::string sib;
        for (int b2 = 0; b2 < grid; ++b2) {
          if (b2 == b) continue;
          const rgm::AttnDbg& e = S.hdbg[b2];
          if (cu_key(e) == cu_key(d) && e.t0 < d.t1 && d.t0 < e.t1) {
            char tmp[96];
            snprintf(tmp, sizeof tmp, " sib b%d alloc %08x [%lld,%lld]", b2, e.lds_alloc, (long long)(e.t0 - d.t0), (long long)(e.t1 - d.t0));
            sib += tmp;
          }
        }
        if (printed < 12) {
          ++printed;
          printf("  wrong query rows of the block: %016llx %016llx (bit r = row r)\n", rowmask[0], rowmask[1]);
          if (o.dump) {
            CK(hipMemcpy(S.dcur.data(), S.dump, dump_n * 4, hipMemcpyDeviceToHost));
            for (int wv = 0; wv < 4; ++wv)
              for (int it = 0; it < ATTN_DUMP_ITEMS; ++it) {
                const size_t base = (((size_t)b * 4 + wv) * ATTN_DUMP_ITEMS + it) * 64;
                unsigned long long lanes = 0;
                for (int ln = 0; ln < 64; ++ln)
                  if (memcmp(&S.dcur[base + ln], &S.dref[base + ln], 4) != 0) lanes |= 1ull << ln;
                if (!lanes) continue;
                int l0 = 0;
                while (!((lanes >> l0) & 1)) ++l0;
                unsigned ua, ub;
                memcpy(&ua, &S.dcur[base + l0], 4);
                memcpy(&ub, &S.dref[base + l0], 4);
                if (it >= 40) {
                  printf("    wave %d: %s differs in lanes %016llx (lane %d: %.9g vs reference %.9g)\n", wv, it == 40 ? "max" : "sum", lanes, l0,
                         S.dcur[base + l0], S.dref[base + l0]);
                  continue;
                }
                printf("    wave %d: Q fragment j=%d %s dword %d differs in lanes %016llx (lane %d: %08x vs reference %08x)\n", wv, it / 8, (it & 4) ? "lo" : "hi",
                       it & 3, lanes, l0, ua, ub);
                if (it == 8 + 1) {   // qh[1] dword 1, low half = element 2 = channel 16 + 8 hh + 2 of query wv*32 + (lane & 31)
                  const int qrow = wv * 32 + (l0 & 31), hh_ = l0 >> 5, ch = 16 + 8 * hh_ + 2;
                  const float* qr = &S.hq[((size_t)n * T + qrow) * 3 * D + h * HD];
                  const float x2 = qr[ch], x3 = qr[ch + 1], c1 = S.hc[qrow * ROT_HALF + ch / 2], s1 = S.hs[qrow * ROT_HALF + ch / 2];
                  const float scl = 1.0f / sqrtf((float)HD) * 1.44269504088896340736f;
                  auto bf = [](float f) {
                    unsigned u;
                    memcpy(&u, &f, 4);
                    u += 0x7fff + ((u >> 16) & 1);
                    return u >> 16;
                  };
                  printf("      that element as bf16: x2*c1 - x3*s1 (right) = %04x; -x3*s1 (x2*c1 lost: c1 read as the OLD contents of its register) = %04x; "
                         "x2*c1 + x3*s1 = %04x; x2*c1 = %04x\n",
                         bf((x2 * c1 - x3 * s1) * scl), bf(-x3 * s1 * scl), bf((x2 * c1 + x3 * s1) * scl), bf(x2 * c1 * scl));
                }
              }
          }
          // does a wrong row equal some OTHER row of the reference (an address / identity mix-up), bit for bit?
          int shown = 0;
          for (int r = 0; r < T && shown < 3; ++r) {
            if (!((rowmask[r >> 6] >> (r & 63)) & 1)) continue;
            ++shown;
            const float* cr = &S.cur[((size_t)n * T + r) * D + h * HD];
            int found = 0;
            for (int b2 = 0; b2 < grid && found < 3; ++b2)
              for (int r2 = 0; r2 < T; ++r2) {
                const float* rr = &S.ref[((size_t)(b2 / HEADS) * T + r2) * D + (b2 % HEADS) * HD];
                if (memcmp(cr, rr, HD * 4) == 0) {
                  printf("    row %d of block %d == reference row %d of block %d\n", r, b, r2, b2);
                  ++found;
                }
              }
            if (!found) {
              const float* er = &S.ref[((size_t)n * T + r) * D + h * HD];
              printf("    row %d matches no reference row; cur/ref ch0..7:", r);
              for (int c = 0; c < 8; ++c) printf(" %.4f/%.4f", cr[c], er[c]);
              printf("\n");
            }
          }
          printf("  [%s] launch %d block %d (n %d head %d): %d wrong elems (%d NaN; lanes 0-31: %d, lanes 32-63: %d), max|err| %.3g, cur/ref over the block %.4f .. %.4f, "
                 "largest spread of cur/ref inside one query row %.2e, query tiles mask %x, hw_id %08x xcc %x cu %04x lds_alloc %08x arrivals %u dur %lld ticks;%s\n",
                 o.name, l, b, n, h, nw, nn, nw_h[0], nw_h[1], mx, rmin, rmax, row_spread, tiles, d.hw_id, d.xcc_id & 0xf, cu_key(d), d.lds_alloc, d.arrivals,
                 (long long)(d.t1 - d.t0), sib.c_str());
        }
      }
    }
    if (any) ++wrong_launches;
  }
  printf("[%s] threads %d lds %zu: %d of %d launches wrong, %d wrong workgroups (%d with NaN) of %d; nonzero-LDS-base workgroups %d; wrong with base!=0 %d, base==0 %d",
         o.name, o.threads, o.lds, wrong_launches, launches, wrong_blocks, nan_blocks, total_blocks, upper_blocks, wrong_upper, wrong_lower);
  if (o.count) printf("; barrier saw fewer arrivals than waves in %d workgroups", short_arrivals);
  printf("\n   LDS_ALLOC histogram:");
  for (auto& kv : alloc_hist) printf(" %08x:%d(w%d)", kv.first, kv.second, alloc_wrong.count(kv.first) ? alloc_wrong[kv.first] : 0);
  printf("\n");
  fflush(stdout);
  CK(hipStreamDestroy(sa));
  CK(hipStreamDestroy(sb));
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 40;
  if (argc > 2) N = atoi(argv[2]);
  const char* only = argc > 3 ? argv[3] : nullptr;
  Setup S;
  const size_t nq = (size_t)N * T * 3 * D, no = (size_t)N * T * D;
  std::vector<float> hq(nq), hc(T * ROT_HALF), hs(T * ROT_HALF);
  unsigned long long seed = 12345;
  for (auto& v : hq) v = 1.5f * frand(seed);
  for (int t = 0; t < T; ++t)
    for (int i = 0; i < ROT_HALF; ++i) {
      const double ang = t * pow(10000.0, -(double)(2 * i) / (2 * ROT_HALF));
      hc[t * ROT_HALF + i] = (float)cos(ang);
      hs[t * ROT_HALF + i] = (float)sin(ang);
    }
  CK(hipMalloc(&S.qkv, nq * 4));
  CK(hipMalloc(&S.out, no * 4));
  CK(hipMalloc(&S.cosd, hc.size() * 4));
  CK(hipMalloc(&S.sind, hs.size() * 4));
  CK(hipMalloc(&S.dbg, sizeof(rgm::AttnDbg) * N * HEADS));
  CK(hipMalloc(&S.cnt, sizeof(int) * N * HEADS));
  CK(hipMalloc(&S.where, 4096));
  CK(hipMalloc(&S.dump, (size_t)N * HEADS * 4 * ATTN_DUMP_ITEMS * 64 * 4));
  CK(hipHostMalloc(&S.flag_host, 4, hipHostMallocMapped));
  CK(hipMemcpy(S.qkv, hq.data(), nq * 4, hipMemcpyHostToDevice));
  S.hq = hq;
  S.hc = hc;
  S.hs = hs;
  CK(hipMemcpy(S.cosd, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(S.sind, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  S.ref.resize(no);
  S.cur.resize(no);
  S.hdbg.resize(N * HEADS);
  CK(hipMemcpyToSymbol(HIP_SYMBOL(rgm::g_attn_dbg), &S.dbg, sizeof(S.dbg)));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(rgm::rotary_attention_x3_kernel<HD, NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_holder_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

  const size_t natural = (size_t)NKT * 32 * (80 * 4 + 16) + (size_t)HD * (NKT * 32 * 4 + 16);   // 81024
  const size_t guard = 80 * 1024 + 512;
  // reference: one workgroup per CU (the shipped guard), checked on a few workgroups against an fp64 evaluation
  launch_attn(S, 512, guard, 0);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(S.ref.data(), S.out, no * 4, hipMemcpyDeviceToHost));
  {
    double worst = 0;
    for (int b : {0, 5 * HEADS + 3, N * HEADS - 1}) {
      const int n = b / HEADS, h = b % HEADS;
      std::vector<double> q(T * HD), k(T * HD), v(T * HD);
      for (int t = 0; t < T; ++t)
        for (int c = 0; c < HD; ++c) {
          const float* row = &hq[((size_t)n * T + t) * 3 * D + h * HD];
          auto rot = [&](const float* p) {
            if (c >= 2 * ROT_HALF) return (double)p[c];
            const int i = c >> 1;
            const float cs = hc[t * ROT_HALF + i], sn = hs[t * ROT_HALF + i];
            return (c & 1) ? (double)(p[c] * cs + p[c - 1] * sn) : (double)(p[c] * cs - p[c + 1] * sn);
          };
          q[t * HD + c] = rot(row);
          k[t * HD + c] = rot(row + D);
          v[t * HD + c] = row[2 * D + c];
        }
      for (int t = 0; t < T; ++t) {
        std::vector<double> s(T);
        double m = -1e300, sum = 0;
        for (int j = 0; j < T; ++j) {
          double a = 0;
          for (int c = 0; c < HD; ++c) a += q[t * HD + c] * k[j * HD + c];
          s[j] = a / sqrt((double)HD);
          m = fmax(m, s[j]);
        }
        for (int j = 0; j < T; ++j) sum += (s[j] = exp(s[j] - m));
        for (int c = 0; c < HD; ++c) {
          double a = 0;
          for (int j = 0; j < T; ++j) a += s[j] * v[j * HD + c];
          worst = fmax(worst, fabs(a / sum - S.ref[((size_t)n * T + t) * D + h * HD + c]));
        }
      }
    }
    printf("reference launch (one workgroup per CU) vs fp64 on 3 workgroups: max abs err %.3g\n", worst);
  }
  auto with_dump = [](Opts o) {
    o.dump = true;
    return o;
  };
  std::vector<Opts> exps = {
      {"guard512", 512, guard},                                               // the shipped launch: one workgroup per CU
      {"guard256", 256, guard},
      {"two512", 512, natural},                                               // round 3's configuration: the second workgroup moves in behind waves 4-7
      {"two512+count", 512, natural, false, true},
      {"two256", 256, natural},                                               // two workgroups per CU from the first cycle, in lockstep: the reproducer
      {"two256+nanfill", 256, natural, true},                                 // stale LDS contents?  (no NaN ever comes out)
      {"two256+count", 256, natural, false, true},                            // early barrier release?  (every workgroup sees all its waves)
      {"upper256 (holder below, no sibling)", 256, natural, false, false, true},   // placement in the upper LDS half alone?  (never wrong)
      {"upper512 (holder below, no sibling)", 512, natural, false, false, true},
#ifdef RGM_ATTN_HAZARD_DUMP
      with_dump({"dump two256", 256, natural}),                               // which Q fragment dword is wrong, and what it holds
#endif
  };

  for (auto& o : exps)
    if (!only || strstr(o.name, only)) experiment(S, o, launches);
  return 0;
}
