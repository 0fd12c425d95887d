// libsoda_hip.so -- run-time of the SODA HIP back end for MI355X (gfx950).
//
// Thin and stateless per call apart from the handles it hands out: a module
// (one code object), a plan (program + kernels + scratch memory).  The stencil
// arithmetic lives in the generated kernels (soda_hip/codegen/kernel_*.py); this
// file decides which kernel runs on which box with which buffers, launches it on
// the caller's stream and times it with hipEvents.
//
// Reference counterparts are cited per function in include/soda_hip.h.
#include "soda_hip.h"

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(code, call)                                                   \
  do {                                                                        \
    hipError_t e_ = (call);                                                   \
    if (e_ != hipSuccess)                                                     \
      return fail((code), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                        \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Experiment knobs (chunk length, XCD super-tile shape, schedule trace) are read
// from the environment ONLY when SODA_HIP_TUNING=1 is set as well: tools/ set it,
// nothing else does, so a stray variable cannot change how a production run is
// scheduled.  None of them can change results, only placement and chunking.
const char* tuning_env(const char* name) {
  const char* on = getenv("SODA_HIP_TUNING");
  return (on && on[0] == '1') ? getenv(name) : nullptr;
}

struct Box {
  int32_t lo[SODA_HIP_MAX_DIMS];  // <= 0
  int32_t hi[SODA_HIP_MAX_DIMS];  // >= 0
  bool set;
};

}  // namespace

struct soda_hip_module {
  hipModule_t mod = nullptr;
  std::vector<char> image;
  std::string meta;
};

struct soda_hip_plan {
  soda_hip_module* module = nullptr;
  soda_hip_program prog{};
  std::vector<soda_hip_kernel> kernels;
  std::vector<hipFunction_t> funcs;
  std::vector<int> resident_blocks;  // per kernel: workgroups the chip holds at once
  std::vector<int> static_lds;       // per kernel: bytes of LDS the code object declares
  int cus = 256;                     // compute units of the device the plan lives on
  int64_t lds_per_cu = 160 * 1024;   // LDS of one CU (gfx950: 160 KiB)
  int max_depth = 0;
  int chunk_rows_override = 0;       // SODA_HIP_CHUNK_ROWS, for tuning
  // shortest chunk the launcher considers: small grids need many short chunks to
  // reach every CU (jacobi3d 128^3, depth 4: 67 us per launch with 32-plane
  // chunks)
  int chunk_rows_min = 8;
  int wgs_per_cu_cap = 0;            // SODA_HIP_WGS_PER_CU, for tuning (see make_launch)
  // soda_hip_plan_set_out_final_only: `out` is written by the LAST launch of a sweep
  // only; the launches before it alternate between scratch and scratch_b
  bool out_final_only = false;
  std::vector<void*> scratch_b;      // second partner per output (out_final_only)
  std::vector<size_t> scratch_b_bytes;
  // scratch: [0, n_outputs) ping-pong partner of the outputs,
  // then one per non-output stage (only used by per-stage kernels)
  std::vector<void*> scratch;
  std::vector<size_t> scratch_bytes;
  // composed boxes per iteration per stage, grown on demand
  std::vector<std::vector<Box>> boxes;
  std::vector<Box> feed;
  // XCD super-tile shape chosen per (kernel, tiles along x, y, chunks): the search
  // walks every super-tile and a sweep's launches mostly repeat a few grids
  mutable std::map<std::array<int64_t, 4>, std::pair<int, int>> xcd_shape;
  // soda_hip_plan_tune: the split of `iterate` (fused depths, deepest first) that ran
  // fastest on this device for arrays of these extents, keyed by dims + iterate (the
  // margins of a resumed or sharded run move the boxes by a few cells, not the
  // ranking); where an entry exists build_schedule uses it instead of its own split
  std::map<std::array<int64_t, 5>, std::vector<int>> tuned_split;
  // soda_hip_plan_tune, streaming launches: the (chunk length, workgroups per CU) that
  // ran fastest on THIS device for a kernel on a box of these extents, keyed by kernel
  // index + box extents; make_launch uses it instead of the kernel's calibration record
  // (stream_chunk / stream_wgs_per_cu were measured on one box of one round)
  std::map<std::array<int64_t, 5>, std::array<int, 2>> tuned_stream;
  // while tuning: the modelled price of kernels of this depth is scaled by this
  // factor (how the candidate splits are generated); 0 = no bias
  int bias_depth = 0;
  double bias = 1.0;
  bool tuning = false;               // candidates are being timed: ignore tuned_split
  // soda_hip_run_slab, bands-first order: the exchange runs on a stream the plan owns
  hipStream_t side = nullptr;
  hipEvent_t ev_main = nullptr, ev_landed = nullptr;
  // soda_hip_clock_probe_start / _finish
  hipFunction_t probe = nullptr;
  void* probe_buf = nullptr;
  bool probe_running = false;
  hipEvent_t probe_t0 = nullptr, probe_t1 = nullptr;
};

namespace {

int n_tensors(const soda_hip_program& p) { return p.n_inputs + p.n_stages; }

bool is_output_tensor(const soda_hip_program& p, int t) {
  for (int j = 0; j < p.n_outputs; ++j)
    if (p.output_tensor[j] == t) return true;
  return false;
}

// Composed read windows back to the original inputs, one iteration at a time
// (reference core.py:794-835 on bounding boxes; output j feeds input j,
// core.py:342-360).
void grow_boxes(soda_hip_plan* plan, int iterations) {
  const soda_hip_program& p = plan->prog;
  const int nt = n_tensors(p);
  if (plan->boxes.empty()) {
    plan->feed.assign(p.n_inputs, Box{});
    for (auto& b : plan->feed) b.set = true;
  }
  while ((int)plan->boxes.size() < iterations) {
    std::vector<Box> cur(nt, Box{});
    for (int i = 0; i < p.n_inputs; ++i) cur[i] = plan->feed[i];
    for (int s = 0; s < p.n_stages; ++s) {
      const int t = p.n_inputs + s;
      Box acc{};
      for (int w = 0; w < p.n_windows; ++w) {
        const soda_hip_window& win = p.window[w];
        if (win.stage != t) continue;
        const Box& par = cur[win.parent];
        for (int d = 0; d < p.dim; ++d) {
          const int32_t lo = par.lo[d] + win.lo[d], hi = par.hi[d] + win.hi[d];
          acc.lo[d] = acc.set ? std::min(acc.lo[d], lo) : lo;
          acc.hi[d] = acc.set ? std::max(acc.hi[d], hi) : hi;
        }
        acc.set = true;
      }
      // the cell itself must lie inside the array: boxes contain the origin
      for (int d = 0; d < p.dim; ++d) {
        acc.lo[d] = std::min<int32_t>(acc.lo[d], 0);
        acc.hi[d] = std::max<int32_t>(acc.hi[d], 0);
      }
      cur[t] = acc;
    }
    if (p.n_inputs == p.n_outputs)
      for (int j = 0; j < p.n_inputs; ++j) plan->feed[j] = cur[p.output_tensor[j]];
    plan->boxes.push_back(cur);
  }
}

// hull over the outputs after `iterations` iterations, as positive margins
void output_margins(soda_hip_plan* plan, int iterations, int32_t* lo, int32_t* hi) {
  const soda_hip_program& p = plan->prog;
  for (int d = 0; d < SODA_HIP_MAX_DIMS; ++d) lo[d] = hi[d] = 0;
  if (iterations <= 0) return;
  grow_boxes(plan, iterations);
  const std::vector<Box>& b = plan->boxes[iterations - 1];
  for (int j = 0; j < p.n_outputs; ++j) {
    const Box& o = b[p.output_tensor[j]];
    for (int d = 0; d < p.dim; ++d) {
      lo[d] = std::max(lo[d], -o.lo[d]);
      hi[d] = std::max(hi[d], o.hi[d]);
    }
  }
}

int ensure_scratch(soda_hip_plan* plan, const int64_t* dims, bool need_locals,
                   bool need_pingpong, hipStream_t stream, bool need_second = false) {
  const soda_hip_program& p = plan->prog;
  size_t cells = 1;
  for (int d = 0; d < p.dim; ++d) cells *= (size_t)dims[d];
  const int n_slots = p.n_outputs + p.n_stages;
  if ((int)plan->scratch.size() != n_slots) {
    plan->scratch.assign(n_slots, nullptr);
    plan->scratch_bytes.assign(n_slots, 0);
  }
  auto want = [&](int slot, size_t bytes) -> int {
    if (plan->scratch_bytes[slot] >= bytes) return 0;
    if (plan->scratch[slot]) {
      HIP_TRY(SODA_HIP_ERR_DEVICE_FREE, hipFree(plan->scratch[slot]));
      plan->scratch[slot] = nullptr;
      plan->scratch_bytes[slot] = 0;
    }
    HIP_TRY(SODA_HIP_ERR_DEVICE_MALLOC, hipMalloc(&plan->scratch[slot], bytes));
    // unspecified cells must at least be readable, finite-ish garbage: zero.
    // On the sweep's own stream: a null-stream memset is not ordered before
    // kernels on a non-blocking stream and could clobber their results.
    HIP_TRY(SODA_HIP_ERR_DEVICE_RUN,
            hipMemsetAsync(plan->scratch[slot], 0, bytes, stream));
    plan->scratch_bytes[slot] = bytes;
    return 0;
  };
  if (need_pingpong)
    for (int j = 0; j < p.n_outputs; ++j) {
      int rc = want(j, cells * p.elem_size[p.output_tensor[j]]);
      if (rc) return rc;
    }
  if (need_second) {
    if ((int)plan->scratch_b.size() != p.n_outputs) {
      plan->scratch_b.assign(p.n_outputs, nullptr);
      plan->scratch_b_bytes.assign(p.n_outputs, 0);
    }
    for (int j = 0; j < p.n_outputs; ++j) {
      const size_t bytes = cells * p.elem_size[p.output_tensor[j]];
      if (plan->scratch_b_bytes[j] >= bytes) continue;
      if (plan->scratch_b[j]) {
        HIP_TRY(SODA_HIP_ERR_DEVICE_FREE, hipFree(plan->scratch_b[j]));
        plan->scratch_b[j] = nullptr;
        plan->scratch_b_bytes[j] = 0;
      }
      HIP_TRY(SODA_HIP_ERR_DEVICE_MALLOC, hipMalloc(&plan->scratch_b[j], bytes));
      HIP_TRY(SODA_HIP_ERR_DEVICE_RUN, hipMemsetAsync(plan->scratch_b[j], 0, bytes, stream));
      plan->scratch_b_bytes[j] = bytes;
    }
  }
  if (need_locals)
    for (int s = 0; s < p.n_stages; ++s) {
      const int t = p.n_inputs + s;
      if (is_output_tensor(p, t)) continue;
      int rc = want(p.n_outputs + s, cells * p.elem_size[t]);
      if (rc) return rc;
    }
  return 0;
}

struct Launch {
  int kernel;
  soda_hip_args args;
  unsigned grid[3];
  double est_us;   // modelled duration (0 = the kernel carries no cost figures)
  unsigned lds_bytes = 0;   // dynamic LDS asked for only to cap the workgroups per CU
  long long rounds = 0;     // streaming kernels: chip-fulls of workgroups the price assumes
  long long resident = 0;   // ... and the workgroups one chip-full is (after any cap)
};

// Cost model of a streaming launch (what the scheduler compares depths with; it
// never has to be right in absolute terms).  A workgroup walks `steps` rows or
// planes; with R workgroups resident per CU one step of all of them takes
//   max( R * step_valu / 4 SIMDs / (clock * issue efficiency),
//        R * CUs * step_bytes / HBM rate this access pattern reaches ).
// Constants measured on MI355X with the jacobi2d kernels of every depth
// (tools/chunk_sweep.py, 16384^2): step_valu carries the arithmetic plus a fixed
// cost per streamed row (barrier, ring, hand-offs; kernel.py: annotate_cost) and
// is issued at the ~2.0 GHz the chip holds under that load; the shallow kernels
// move 4.5-4.7 TB/s.  Modelled vs measured us per step of a full chip: depth 12
// 0.91 / 0.89, 16 0.96 / 0.98, 20 0.84 / 0.84, 24 0.96 / 0.96.
const double kModelValuHz = 2.0e9;
const double kModelHbmBytesPerSec = 4.6e12;
const double kModelLaunchUs = 2.0;

// `footprint` = bytes of the arrays the launch streams (inputs + outputs of its box).
// With calibration figures in the descriptor (soda_hip_kernel.step_ns_*,
// stream_gbps): the kernel's own step time at the occupancy this grid reaches,
// interpolated between one workgroup per CU and a full chip, or - on arrays beyond
// the Infinity Cache - the time its HBM rate allows, whichever is longer.
// The HBM term fades in between arrays that live in the 256 MiB Infinity Cache and
// arrays several times its size (jacobi3d, one-level-per-wavefront kernel: 1.63 us per
// step at 304^3 = 215 MiB in + out, the step time of a cache-resident array; 2.03 us at
// 400^3 = 488 MiB; 2.41 us at 512^3).
// The two footprints are part of the kernel's calibration record (soda_hip_kernel.
// fade_lo_mib / fade_hi_mib, tools/calibrate.py); these are the defaults of kernels
// that carry none.
const double kCacheResidentMiB = 128.0;
const double kStreamingMiB = 512.0;
// beyond this a launch's box does not fit the 256 MiB Infinity Cache (the kernels' own
// non-temporal paths switch at the same figure: kernel_common.NT_STREAMING_BYTES)
const double kBeyondCacheBytes = 288.0 * 1024 * 1024;
const int64_t kMaxGridYZ = 65535;      // workgroups along grid.y / grid.z

// `resident` = workgroups of this launch the chip holds at once (the kernel's occupancy,
// or less under a cap on workgroups per CU; 0 = the kernel's occupancy)
double step_seconds(const soda_hip_plan* plan, int k, double blocks, double footprint = 0,
                    double resident = 0) {
  const soda_hip_kernel& desc = plan->kernels[k];
  const double cus = std::max(1, plan->cus);
  const double full = std::max(1, plan->resident_blocks[k]) / cus;
  const double held = resident > 0 ? std::min(full, resident / cus) : full;
  // a grid smaller than the chip holds: fewer workgroups share each CU
  const double per_cu = std::min(held, std::max(1.0, blocks / cus));
  if (desc.step_ns_full > 0 && desc.step_ns_one > 0) {
    const double share = full > 1 ? (per_cu - 1) / (full - 1) : 1.0;
    double t = (desc.step_ns_one + (desc.step_ns_full - desc.step_ns_one) * share) * 1e-9;
    const double mib = 1024.0 * 1024.0;
    const double fade_lo = (desc.fade_lo_mib > 0 ? desc.fade_lo_mib : kCacheResidentMiB) * mib;
    const double fade_hi = std::max(fade_lo + mib,
        (desc.fade_hi_mib > 0 ? desc.fade_hi_mib : kStreamingMiB) * mib);
    if (desc.stream_gbps > 0 && desc.step_bytes > 0 && footprint > fade_lo) {
      const double weight = std::min(1.0, (footprint - fade_lo) / (fade_hi - fade_lo));
      t = std::max(t, weight * std::min(blocks, held * cus) * desc.step_bytes /
                          (desc.stream_gbps * 1e9));
    }
    return t;
  }
  if (desc.step_valu <= 0 && desc.step_bytes <= 0) return 0;
  const double valu = per_cu * desc.step_valu / 4.0 / kModelValuHz;
  const double hbm = per_cu * cus * desc.step_bytes / kModelHbmBytesPerSec;
  return std::max(valu, hbm);
}

// bytes a launch streams: its box, every input and output
double footprint_of(const soda_hip_plan* plan, const soda_hip_args& args) {
  const soda_hip_program& p = plan->prog;
  double cells = 1;
  for (int e = 0; e < p.dim; ++e) cells *= (double)(args.box_hi[e] - args.box_lo[e]);
  double footprint = 0;
  for (int j = 0; j < p.n_inputs; ++j) footprint += cells * p.elem_size[j];
  for (int j = 0; j < p.n_outputs; ++j) footprint += cells * p.elem_size[p.output_tensor[j]];
  return footprint;
}

// workgroups per CU a streaming launch of kernel k is capped at (0 = no cap):
// soda_hip_kernel.stream_wgs_per_cu for boxes beyond the Infinity Cache
// (tuning: SODA_HIP_WGS_PER_CU = N for every streaming kernel, -1 = never)
int streaming_cap(const soda_hip_plan* plan, int k, double footprint) {
  if (plan->wgs_per_cu_cap != 0) return std::max(0, plan->wgs_per_cu_cap);
  if (footprint <= kBeyondCacheBytes) return 0;
  return std::max(0, (int)plan->kernels[k].stream_wgs_per_cu);
}

std::array<int64_t, 5> stream_key(const soda_hip_plan* plan, int k, const soda_hip_args& args) {
  std::array<int64_t, 5> key = {k, 1, 1, 1, 1};
  for (int d = 0; d < plan->prog.dim && d < 4; ++d) key[1 + d] = args.box_hi[d] - args.box_lo[d];
  return key;
}

int make_launch(const soda_hip_plan* plan, int k, const soda_hip_args& args,
                Launch* out, bool* empty) {
  const soda_hip_kernel& desc = plan->kernels[k];
  const int dim = plan->prog.dim;
  out->kernel = k;
  out->args = args;
  out->est_us = 0;
  out->lds_bytes = 0;
  *empty = false;
  // never launch a box that sticks out of the array
  for (int d = 0; d < dim; ++d)
    if (args.box_hi[d] > args.box_lo[d] &&
        (args.box_lo[d] < 0 || args.box_hi[d] > args.dims[d]))
      return fail(SODA_HIP_ERR_OUT_OF_BOUNDS,
                  "kernel %s: box [%lld, %lld) outside dimension %d of extent %lld",
                  desc.name, (long long)args.box_lo[d], (long long)args.box_hi[d], d,
                  (long long)args.dims[d]);
  for (int d = 0; d < 3; ++d) out->grid[d] = 1;
  int64_t edge_origin = -1;      // soda_hip_kernel.edge_slack: where the tiles start along x
  if (dim > 3 && desc.kind != SODA_HIP_KERNEL_STAGE)
    return fail(SODA_HIP_ERR_INTERNAL, "kernel %s: only per-stage kernels take 4-D boxes",
                desc.name);
  bool fold_rows = dim > 3;   // a 4-D box always goes as folded rows
  for (int d = 0; d < dim; ++d) {
    int64_t extent = args.box_hi[d] - args.box_lo[d];
    if (extent <= 0) { *empty = true; return 0; }
    if (d == 0 && desc.origin_align > 1)   // tiles start at an aligned column
      extent += args.box_lo[0] % desc.origin_align;
    if (desc.tile[d] <= 0)
      return fail(SODA_HIP_ERR_INTERNAL, "kernel %s has tile[%d]=%d", desc.name, d,
                  desc.tile[d]);
    int64_t tile = desc.tile[d];
    if (d == dim - 1 && desc.fill_rows > 0 && dim >= 2) {
      // Streaming kernel: every workgroup walks `chunk + fill_rows` rows of the
      // outer dimension.  Pick the chunk length that minimises
      //   rounds(chunk) * (chunk + fill_rows),
      // rounds = ceil(workgroups / workgroups resident on the chip): a grid
      // that is 2.4 chip-fulls costs 3, so aim for whole rounds.
      int64_t inner = 1;
      for (int e = 0; e < dim - 1; ++e) inner *= out->grid[e];
      int64_t resident = std::max(1, plan->resident_blocks[k]);
      // A cap on the workgroups a CU holds at once (streaming launches of the
      // memory-bound kernels: fewer wavefronts walking longer chunks keep the set of
      // DRAM pages the chip touches at a time small - tools/copyceil.hip): enforced
      // with dynamic LDS the kernel never uses, 160 KiB / (cap + 1) + 1 KiB each
      int cap = streaming_cap(plan, k, footprint_of(plan, args));
      int64_t tuned_chunk = 0;
      if (desc.stream_chunk > 0 && footprint_of(plan, args) > kBeyondCacheBytes &&
          plan->chunk_rows_override == 0 && plan->wgs_per_cu_cap == 0) {
        const auto tuned = plan->tuned_stream.find(stream_key(plan, k, args));
        if (tuned != plan->tuned_stream.end()) {
          tuned_chunk = tuned->second[0];
          cap = tuned->second[1];
        }
      }
      if (cap > 0 && resident > (int64_t)cap * plan->cus) {
        // each workgroup must take more than 1 / (cap + 1) of the CU's LDS and at most
        // 1 / cap of it, its static LDS included; a cap the padding cannot realise (the
        // static part alone already excludes `cap` workgroups) is not applied
        const int64_t lds_cu = plan->lds_per_cu, fixed = plan->static_lds[k];
        const int64_t granule = 1024;
        int64_t total = lds_cu / (cap + 1) / granule * granule + granule;   // > lds_cu / (cap + 1)
        total = std::max(total, (fixed + granule - 1) / granule * granule);
        if (total * cap <= lds_cu) {
          resident = (int64_t)cap * plan->cus;
          out->lds_bytes = (unsigned)std::max<int64_t>(0, total - fixed);
        }
      }
      int64_t best = tile, best_cost = -1;
      const double footprint = footprint_of(plan, args);
      const int64_t shortest = plan->chunk_rows_min;   // 8; SODA_HIP_CHUNK_MIN
      for (int64_t chunk = shortest;
           chunk <= std::max<int64_t>(shortest, std::min<int64_t>(extent, 4096));
           chunk += 4) {
        const int64_t blocks = inner * ((extent + chunk - 1) / chunk);
        const int64_t rounds = (blocks + resident - 1) / resident;
        const int64_t cost = rounds * (chunk + desc.fill_rows);
        // among equal step counts the LONGEST chunk: fewer workgroups, fewer fill rows
        // fetched (jacobi3d box 504^3: 5 chunks of 104 planes in one round and 11 of
        // 48 in two both walk 112 steps; the long ones read 8 % less).  (Round 3 also
        // measured the chunk by its PRICED time - cfg4 +16 %, cfg2 +7 % - and the shortest
        // chunk on ties - cfg5 +3 %: docs/DESIGN_HISTORY.md 4.3; both switches are gone.)
        if (best_cost < 0 || cost <= best_cost) {
          best_cost = cost;
          best = chunk;
        }
      }
      // the kernel's measured chunk for boxes beyond the cache (soda_hip_kernel.
      // stream_chunk): short chunks in dispatch order keep the rows in flight together
      // (tuning: SODA_HIP_CHUNK_ROWS = N forces N, -1 the rule above whatever the kernel says)
      if (desc.stream_chunk > 0 && footprint > kBeyondCacheBytes &&
          plan->chunk_rows_override == 0)
        best = std::max<int64_t>(1, std::min<int64_t>(
            tuned_chunk > 0 ? tuned_chunk : desc.stream_chunk, extent));
      if (plan->chunk_rows_override > 0) best = plan->chunk_rows_override;
      // ... but never so short that the chunks outnumber what one grid dimension takes
      // (a 256 x 1M box in chunks of 8 rows would be 125 000 workgroups along y)
      best = std::max<int64_t>(best, (extent + kMaxGridYZ - 1) / kMaxGridYZ);
      tile = best;
      out->args.param[0] = best;
      const double blocks = (double)inner * (double)((extent + best - 1) / best);
      const double rounds = std::ceil(blocks / (double)resident);
      out->rounds = (long long)rounds;
      out->est_us = kModelLaunchUs + rounds * (double)(best + desc.fill_rows) *
                                         step_seconds(plan, k, blocks, footprint,
                                                      (double)resident) * 1e6;
      out->resident = (long long)resident;
    }
    int64_t g = (extent + tile - 1) / tile;
    if (d == 0 && desc.edge_slack > 0 && desc.origin_align > 1 && desc.xcd_tiles < 0 && dim == 3) {
      // the first and the last tile of a row store the columns the alignment left over
      // (include/soda_hip.h: edge_slack): tiles start up to `slack` columns inside the box
      const int64_t slack = desc.edge_slack, lo = args.box_lo[0], hi = args.box_hi[0];
      int64_t x0 = (lo + slack) - (lo + slack) % desc.origin_align;
      if (x0 >= hi) x0 = lo - lo % desc.origin_align;     // a box narrower than the shift
      int64_t nx = std::max<int64_t>(1, (hi - x0 - slack + tile - 1) / tile);
      if (nx == 1 && hi > x0 + tile) {      // one tile cannot stretch both ways
        x0 = lo - lo % desc.origin_align;
        nx = std::max<int64_t>(1, (hi - x0 - slack + tile - 1) / tile);
      }
      edge_origin = x0;
      g = nx;
    }
    if (d > 0 && desc.kind == SODA_HIP_KERNEL_STAGE && (g > 65535 || dim > 3)) {
      // Per-stage kernels take one row (plane) per workgroup; past the 65535
      // limit of grid.y / grid.z - and for every 4-D box - the rows and planes are
      // folded into one index spread over grid.y x grid.z (kernel_stage.py;
      // param[0] = 1 tells a 3-D kernel so).  Done after the loop, once every
      // extent is known.
      fold_rows = true;
      g = 1;
    }
    if (g > (d == 0 ? 2147483647LL : 65535LL))
      return fail(SODA_HIP_ERR_EXTENTS_TOO_LARGE,
                  "grid dimension %d of kernel %s would be %lld", d, desc.name,
                  (long long)g);
    if (d < 3) out->grid[d] = (unsigned)g;
  }
  if (fold_rows) {
    int64_t rows = 1;
    for (int d = 1; d < dim; ++d) rows *= args.box_hi[d] - args.box_lo[d];
    const int64_t gy = std::min<int64_t>(rows, 65535), gz = (rows + gy - 1) / gy;
    if (gz > 65535)
      return fail(SODA_HIP_ERR_EXTENTS_TOO_LARGE, "kernel %s: %lld rows", desc.name,
                  (long long)rows);
    out->grid[1] = (unsigned)gy;
    out->grid[2] = (unsigned)gz;
    out->args.param[0] = 1;
  }
  if (desc.xcd_tiles < 0 && dim == 3) {
    // Runs (kernel_stream3d_blk.py, xcd_runs): XCD x (= workgroup id mod 8) takes the
    // tiles [x P, (x + 1) P) of the x-fastest order, P = ceil(tiles / 8), so that a
    // tile's x- and y-neighbours stream beside it on the same L2.
    const int64_t gx = out->grid[0], gy = out->grid[1], gz = out->grid[2];
    const int64_t per = (gx * gy * gz + 7) / 8;
    if (per * 8 > 2147483647LL || gx > 65535 || gy > 65535)
      return fail(SODA_HIP_ERR_EXTENTS_TOO_LARGE, "grid of kernel %s would be %lld",
                  desc.name, (long long)(per * 8));
    out->args.param[1] = 1 | (1 << 16);
    if (edge_origin >= 0) out->args.param[1] |= edge_origin << 32;
    out->args.param[2] = gx | (gy << 16);
    out->args.param[3] = per;
    out->grid[0] = (unsigned)(per * 8);
    out->grid[1] = out->grid[2] = 1;
  } else if (desc.xcd_tiles && dim == 3) {
    // XCD-aware placement (kernel_stream3d_wp.py, xcd_tiles): the plane of
    // gx x gy tiles is cut into super-tiles of SX x SY tiles whose workgroups run
    // together on one XCD and share its L2.  Pick the shape that fetches least:
    // padding (tiles beyond the edge) x halo and cache-line slack amortised over
    // the super-tile.
    const int64_t gx = out->grid[0], gy = out->grid[1], gz = out->grid[2];
    const double w = desc.tile[0], r = desc.tile[1];
    const double line = 128.0 / std::max(1, plan->prog.elem_size[0]);
    const double hx = std::max(0, desc.min_extent[0] - desc.tile[0]) + 0.75 * line;
    const double hy = std::max(0, desc.min_extent[1] - desc.tile[1]);
    // Which shapes are eligible (jacobi3d, depth-4 wave-pipelined kernel, one call
    // each; `ids` = workgroup ids launched, padding included):
    //   512^3, 9 x 21 x 4 = 756 tiles on 768 slots: plain deal 378 us, 3 x 1 338,
    //     1 x 3 354, 2 x 1 (840 ids) 435, 1 x 2 (792 ids) 450, 3 x 7 (105 tiles on
    //     four XCDs, 84 on the others) 500
    //   440^3, 8 x 18 x 5 = 720: plain 223, 2 x 1 199, 4 x 1 189, 1 x 2 / 1 x 3 223
    //   392^3, 7 x 16 x 6 = 672: plain 162, 2 x 1 / 4 x 1 (768 ids) 150 / 140
    //   344^3, 6 x 14 x 9 = 756: plain 104, 3 x 1 97, 2 x 2 96, 4 x 1 (1008 ids) 140
    //   264^3, 5 x 11 x 13 = 715: plain 63, 2 x 1 / 3 x 1 (864 ids) 83 / 77;
    //     5 x 11 x 8 = 440: plain 56, 3 x 1 (66 tiles on the even XCDs, 44 on the odd) 67
    // So: (1) a partial super-tile is padded with workgroups that exit at once, and
    // that is harmless only while ALL ids fit the chip at once; (2) super-tiles are
    // dealt whole, so the busiest XCD must stay within 3 % of its even share;
    // (3) grouping along x is what pays (neighbours share 128-byte lines), along y
    // hardly at all.
    const int64_t real = gx * gy * gz;
    const int64_t slots = std::max<int64_t>(8, plan->resident_blocks[k] / 8 * 8);
    const int64_t even = (real + 7) / 8;
    const int64_t limit = even + std::max<int64_t>(1, even * 3 / 100);
    int best_sx = 1, best_sy = 1;
    double best = -1;
    // (the kernel names its largest group, soda_hip_kernel.xcd_tiles: 4 for
    // kernel_stream3d_wp, 1 = the plain deal for the block form.)  Groups of 16-24
    // tiles cut the PMC read bytes further (jacobi3d x200: reads 2.5x -> 1.7x the
    // written bytes) but every one measured ran SLOWER (cfg5 6.2 -> 7.0-7.4 ms)
    int max_group = std::max(1, (int)desc.xcd_tiles);   // the kernel's own limit
    if (const char* env = tuning_env("SODA_HIP_XCD_GROUP")) max_group = std::max(1, atoi(env));
    const std::array<int64_t, 4> key = {k, gx, gy, gz};
    const auto known = plan->xcd_shape.find(key);
    const bool cached = known != plan->xcd_shape.end() && !tuning_env("SODA_HIP_XCD_GROUP");
    if (cached) { best_sx = known->second.first; best_sy = known->second.second; }
    for (int sx = 1; sx <= 8 && !cached; ++sx)
      for (int sy = 1; sy <= 8; ++sy) {
        if (sx * sy > max_group) continue;
        const int64_t nsx = (gx + sx - 1) / sx, nsy = (gy + sy - 1) / sy;
        const int64_t ids = (nsx * nsy * gz + 7) / 8 * 8 * sx * sy;
        if (sx * sy > 1 && ids > (real <= slots ? slots : real + real * 3 / 100)) continue;
        // real tiles per XCD: super-tile g -> XCD g % 8; edge super-tiles are partial
        int64_t per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t g = 0; g < nsx * nsy * gz; ++g) {
          const int64_t tx = std::min<int64_t>(sx, gx - (g % nsx) * sx);
          const int64_t ty = std::min<int64_t>(sy, gy - ((g / nsx) % nsy) * sy);
          per_xcd[g % 8] += tx * ty;
        }
        if (sx * sy > 1 && *std::max_element(per_xcd, per_xcd + 8) > limit) continue;
        const double cost = (1 + hx / (std::min<int64_t>(sx, gx) * w)) *
                            (1 + 0.25 * hy / (std::min<int64_t>(sy, gy) * r));
        if (best < 0 || cost < best) { best = cost; best_sx = sx; best_sy = sy; }
      }
    if (!cached && !tuning_env("SODA_HIP_XCD_GROUP"))
      plan->xcd_shape[key] = std::make_pair(best_sx, best_sy);
    if (const char* env = tuning_env("SODA_HIP_XCD_TILES")) {   // tuning: "SX,SY"
      int sx = 0, sy = 0;
      if (sscanf(env, "%d,%d", &sx, &sy) == 2 && sx > 0 && sy > 0) { best_sx = sx; best_sy = sy; }
    }
    if (tuning_env("SODA_HIP_DEBUG"))
      fprintf(stderr, "soda_hip: %s: %lld x %lld x %lld tiles, super-tiles of %d x %d\n",
              desc.name, (long long)gx, (long long)gy, (long long)gz, best_sx, best_sy);
    const int64_t nsx = (gx + best_sx - 1) / best_sx, nsy = (gy + best_sy - 1) / best_sy;
    const int64_t supers = nsx * nsy * gz;
    const int64_t total = (supers + 7) / 8 * 8 * best_sx * best_sy;
    if (total > 2147483647LL || nsx > 65535 || nsy > 65535)
      return fail(SODA_HIP_ERR_EXTENTS_TOO_LARGE, "grid of kernel %s would be %lld",
                  desc.name, (long long)total);
    out->args.param[1] = best_sx | (best_sy << 16);
    out->args.param[2] = nsx | (nsy << 16);
    out->grid[0] = (unsigned)total;
    out->grid[1] = out->grid[2] = 1;
  }
  return 0;
}

int check_box_inside(const soda_hip_plan* plan, const soda_hip_args& a,
                     const int32_t* reach_lo, const int32_t* reach_hi,
                     bool signed_window = false) {
  // every cell a launch may read must be inside the array: the kernels rely on
  // it.  reach_* are margins (>= 0) or, with signed_window, window offsets
  // (lo <= hi, either sign).
  for (int d = 0; d < plan->prog.dim; ++d) {
    if (a.box_hi[d] <= a.box_lo[d]) continue;
    const int64_t first = signed_window ? a.box_lo[d] + reach_lo[d] : a.box_lo[d] - reach_lo[d];
    if (first < 0 || a.box_hi[d] + reach_hi[d] > a.dims[d])
      return fail(SODA_HIP_ERR_OUT_OF_BOUNDS,
                  "launch would read [%lld, %lld) of dimension %d, extent %lld",
                  (long long)first,
                  (long long)(a.box_hi[d] + reach_hi[d]), d, (long long)a.dims[d]);
  }
  return 0;
}

// Builds the launch list of one sweep.
int build_schedule(soda_hip_plan* plan, void* const* in, void* const* out,
                   const int64_t* dims, int iterate, const int32_t* valid_lo,
                   const int32_t* valid_hi, std::vector<Launch>* list,
                   int* max_depth_used, hipStream_t stream, bool dry = false) {
  const soda_hip_program& p = plan->prog;
  if (iterate < 1) return fail(SODA_HIP_ERR_CONSTRAINT, "iterate must be >= 1");
  if (iterate > 1 && p.n_inputs != p.n_outputs)
    return fail(SODA_HIP_ERR_CONSTRAINT,
                "iterate > 1 needs as many outputs as inputs (%d vs %d)",
                p.n_inputs, p.n_outputs);
  // dry: only the launch list is wanted (soda_hip_plan_schedule): no buffers,
  // no scratch allocation
  std::vector<void*> none(SODA_HIP_MAX_IO, nullptr);
  if (dry) in = out = none.data();
  for (int j = 0; j < p.n_inputs && !dry; ++j)
    if (!in[j]) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "input %d is NULL", j);
  for (int j = 0; j < p.n_outputs && !dry; ++j)
    if (!out[j]) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "output %d is NULL", j);
  int32_t vlo[SODA_HIP_MAX_DIMS] = {0, 0, 0, 0}, vhi[SODA_HIP_MAX_DIMS] = {0, 0, 0, 0};
  for (int d = 0; d < p.dim; ++d) {
    if (valid_lo) vlo[d] = valid_lo[d];
    if (valid_hi) vhi[d] = valid_hi[d];
    if (dims[d] <= 0) return fail(SODA_HIP_ERR_CONSTRAINT, "dims[%d] = %lld", d,
                                  (long long)dims[d]);
  }
  grow_boxes(plan, iterate);
  list->clear();
  *max_depth_used = 0;
  if ((int)plan->scratch.size() != p.n_outputs + p.n_stages) {
    // slots exist (as NULL) before anything is allocated: launch lists may be
    // built without touching memory (soda_hip_plan_schedule)
    plan->scratch.assign(p.n_outputs + p.n_stages, nullptr);
    plan->scratch_bytes.assign(p.n_outputs + p.n_stages, 0);
  }

  // fused kernels available?  (single hull box => single-output programs, or
  // outputs that share a window; the printer only emits them when that holds)
  std::vector<int> fused;  // kernel indices sorted by depth descending
  for (size_t k = 0; k < plan->kernels.size(); ++k) {
    const soda_hip_kernel& kd = plan->kernels[k];
    if (kd.kind != SODA_HIP_KERNEL_FUSED ||
        (plan->max_depth > 0 && kd.depth > plan->max_depth))
      continue;
    // kernels without a guarded path: only arrays at least one tile large; the
    // 3-D ones index inside a plane with 32 bits (2-D ones are 64-bit throughout)
    if (kd.min_extent[0] > 0 &&
        (dims[0] < kd.min_extent[0] || (p.dim > 1 && dims[1] < kd.min_extent[1]) ||
         // (a lane that must not store gets byte offset 0xfffffff0 in the plane's
         // buffer resource: the plane must end below that, widest store included)
         (p.dim > 2 && dims[0] * dims[1] >= (int64_t(1) << 30) - 16)))
      continue;
    fused.push_back((int)k);
  }
  std::sort(fused.begin(), fused.end(), [&](int a, int b) {
    return plan->kernels[a].depth > plan->kernels[b].depth;
  });
  bool fused_ok = !fused.empty() && plan->kernels[fused.back()].depth == 1;
  if (plan->max_depth < 0) fused_ok = false;  // force per-stage kernels

  if (fused_ok) {
    // Split of `iterate` into the available depths: the cheapest one under the
    // cost model (make_launch prices every depth on the first box it would run
    // on; boxes shrink slowly, the ranking holds along the sweep), e.g. jacobi2d
    // x100 = 5 x depth 20 rather than 4 x depth 24 + a memory-bound depth-4 tail.
    // Kernels without cost figures: greedy, deepest first.
    auto first_box = [&](int k) {
      soda_hip_args a;
      memset(&a, 0, sizeof a);
      int32_t mlo[SODA_HIP_MAX_DIMS], mhi[SODA_HIP_MAX_DIMS];
      output_margins(plan, plan->kernels[k].depth, mlo, mhi);
      for (int d = 0; d < SODA_HIP_MAX_DIMS; ++d) {
        a.dims[d] = d < p.dim ? dims[d] : 1;
        a.box_lo[d] = d < p.dim ? vlo[d] + mlo[d] : 0;
        a.box_hi[d] = d < p.dim ? dims[d] - vhi[d] - mhi[d] : 1;
      }
      return a;
    };
    std::vector<int> usable;
    std::vector<double> price;
    bool priced = true;
    for (int k : fused) {
      if (plan->kernels[k].depth > iterate) continue;
      Launch l;
      bool empty = false;
      int rc = make_launch(plan, k, first_box(k), &l, &empty);
      if (rc) return rc;
      if (plan->bias_depth == plan->kernels[k].depth) l.est_us *= plan->bias;
      if (!empty && l.est_us <= 0) priced = false;
      usable.push_back(k);
      price.push_back(empty ? kModelLaunchUs : l.est_us);
    }
    if (usable.empty()) return fail(SODA_HIP_ERR_INTERNAL, "no fused kernel of depth 1");
    std::vector<int> seq;
    std::array<int64_t, 5> tune_key;
    for (int d = 0; d < 4; ++d) tune_key[d] = d < p.dim ? dims[d] : 1;
    tune_key[4] = iterate;
    const auto tuned = plan->tuned_split.find(tune_key);
    if (!plan->tuning && tuned != plan->tuned_split.end()) {
      // the split that ran fastest here (soda_hip_plan_tune): depth -> the first
      // usable kernel of that depth (same-depth alternatives are chosen per launch
      // below, as always)
      for (int depth : tuned->second)
        for (int k : usable)
          if (plan->kernels[k].depth == depth) { seq.push_back(k); break; }
      int total = 0;
      for (int k : seq) total += plan->kernels[k].depth;
      if (total != iterate) seq.clear();     // kernels changed since: fall back
    }
    if (!seq.empty()) {
    } else if (priced) {
      std::vector<double> best(iterate + 1, 1e300);
      std::vector<int> pick(iterate + 1, -1);
      best[0] = 0;
      for (int t = 1; t <= iterate; ++t)
        for (size_t i = 0; i < usable.size(); ++i) {
          const int d = plan->kernels[usable[i]].depth;
          // (<: among equal prices the deeper kernel, listed first, wins)
          if (d <= t && best[t - d] + price[i] < best[t]) {
            best[t] = best[t - d] + price[i];
            pick[t] = (int)i;
          }
        }
      for (int t = iterate; t > 0; t -= plan->kernels[usable[pick[t]]].depth) {
        if (pick[t] < 0) return fail(SODA_HIP_ERR_INTERNAL, "no fused kernel of depth 1");
        seq.push_back(usable[pick[t]]);
      }
      std::sort(seq.begin(), seq.end(), [&](int a, int b) {
        return plan->kernels[a].depth > plan->kernels[b].depth;
      });
    } else {
      for (int left = iterate; left > 0;) {
        int pick = -1;
        for (int k : usable)
          if (plan->kernels[k].depth <= left) { pick = k; break; }
        if (pick < 0) return fail(SODA_HIP_ERR_INTERNAL, "no fused kernel of depth 1");
        seq.push_back(pick);
        left -= plan->kernels[pick].depth;
      }
    }
    const int m = (int)seq.size();
    if (tuning_env("SODA_HIP_DEBUG")) {
      fprintf(stderr, "soda_hip: %d iteration(s) =", iterate);
      for (int k : seq) fprintf(stderr, " %d", plan->kernels[k].depth);
      fprintf(stderr, "  (%s;", priced ? "cost model" : "greedy");
      for (size_t i = 0; i < usable.size(); ++i)
        fprintf(stderr, " k%d %.1f us", plan->kernels[usable[i]].depth, price[i]);
      fprintf(stderr, ")\n");
    }
    // out_final_only: the m - 1 launches before the last alternate between the two
    // plan-owned arrays (two of them need the second one)
    const bool second = plan->out_final_only && m > 2;
    if (m > 1 && !dry) {
      int rc = ensure_scratch(plan, dims, false, true, stream, second);
      if (rc) return rc;
    }
    int done = 0;
    std::vector<void*> src(in, in + p.n_inputs);
    for (int i = 0; i < m; ++i) {
      const soda_hip_kernel& desc = plan->kernels[seq[i]];
      // destinations alternate so that the last one is `out` (out_final_only: the
      // others alternate between the plan's two arrays and never touch `out`)
      const bool to_out = plan->out_final_only ? i == m - 1 : ((m - 1 - i) % 2) == 0;
      const bool to_b = plan->out_final_only && second && ((m - 1 - i) % 2) == 0;
      soda_hip_args a;
      memset(&a, 0, sizeof a);
      for (int j = 0; j < p.n_inputs; ++j) a.tensor[j] = src[j];
      std::vector<void*> dst(p.n_outputs);
      for (int j = 0; j < p.n_outputs; ++j) {
        dst[j] = to_out ? out[j] : to_b && !dry ? plan->scratch_b[j] : plan->scratch[j];
        a.tensor[p.output_tensor[j]] = dst[j];
      }
      int32_t mlo[SODA_HIP_MAX_DIMS], mhi[SODA_HIP_MAX_DIMS];
      int32_t plo[SODA_HIP_MAX_DIMS], phi[SODA_HIP_MAX_DIMS];
      output_margins(plan, done, plo, phi);
      output_margins(plan, done + desc.depth, mlo, mhi);
      int32_t reach_lo[SODA_HIP_MAX_DIMS], reach_hi[SODA_HIP_MAX_DIMS];
      for (int d = 0; d < SODA_HIP_MAX_DIMS; ++d) {
        a.dims[d] = d < p.dim ? dims[d] : 1;
        a.box_lo[d] = d < p.dim ? vlo[d] + mlo[d] : 0;
        a.box_hi[d] = d < p.dim ? dims[d] - vhi[d] - mhi[d] : 1;
        reach_lo[d] = mlo[d] - plo[d];
        reach_hi[d] = mhi[d] - phi[d];
      }
      int rc = check_box_inside(plan, a, reach_lo, reach_hi);
      if (rc) return rc;
      Launch l;
      bool empty;
      rc = make_launch(plan, seq[i], a, &l, &empty);
      if (rc) return rc;
      // several kernels of this depth (3-D: the wave-pipelined form with 64 x 32
      // tiles and the block form with 128 x 64 ones): the cheapest on THIS box -
      // large boxes favour the big tiles, small ones the many small ones
      if (!empty && l.est_us > 0) {
        // (tuning: SODA_HIP_PREFER=<suffix> takes the same-depth kernel whose name
        // ends in it whatever the estimates say - tools/ compare kernel forms with it)
        const char* prefer = tuning_env("SODA_HIP_PREFER");
        auto preferred = [&](int k) {
          if (!prefer) return false;
          const size_t n = strlen(plan->kernels[k].name), m = strlen(prefer);
          return n >= m && strcmp(plan->kernels[k].name + n - m, prefer) == 0;
        };
        for (int k : fused) {
          if (k == seq[i] || plan->kernels[k].depth != desc.depth) continue;
          Launch other;
          bool other_empty;
          if (make_launch(plan, k, a, &other, &other_empty) == 0 && !other_empty &&
              other.est_us > 0 &&
              (preferred(k) || (other.est_us < l.est_us && !preferred(l.kernel))))
            l = other;
        }
      }
      if (!empty) list->push_back(l);
      *max_depth_used = std::max(*max_depth_used, (int)desc.depth);
      done += desc.depth;
      if (p.n_inputs == p.n_outputs) src = dst;
    }
    return 0;
  }

  // per-stage kernels: one launch per stage per iteration, intermediates in HBM
  std::vector<int> stage_kernel(p.n_stages, -1);
  for (size_t k = 0; k < plan->kernels.size(); ++k)
    if (plan->kernels[k].kind == SODA_HIP_KERNEL_STAGE) {
      const int s = plan->kernels[k].stage - p.n_inputs;
      if (s >= 0 && s < p.n_stages) stage_kernel[s] = (int)k;
    }
  for (int s = 0; s < p.n_stages; ++s)
    if (stage_kernel[s] < 0)
      return fail(SODA_HIP_ERR_NO_KERNEL, "blob has no kernel for stage %d", s);
  const bool second_st = plan->out_final_only && iterate > 2;
  if (!dry) {
    int rc = ensure_scratch(plan, dims, true, iterate > 1, stream, second_st);
    if (rc) return rc;
  }
  std::vector<void*> src(in, in + p.n_inputs);
  for (int it = 0; it < iterate; ++it) {
    const bool to_out = plan->out_final_only ? it == iterate - 1
                                             : ((iterate - 1 - it) % 2) == 0;
    const bool to_b = second_st && !dry && ((iterate - 1 - it) % 2) == 0;
    soda_hip_args a;
    memset(&a, 0, sizeof a);
    for (int j = 0; j < p.n_inputs; ++j) a.tensor[j] = src[j];
    std::vector<void*> dst(p.n_outputs);
    for (int s = 0; s < p.n_stages; ++s)
      a.tensor[p.n_inputs + s] = plan->scratch[p.n_outputs + s];
    for (int j = 0; j < p.n_outputs; ++j) {
      dst[j] = to_out ? out[j] : to_b ? plan->scratch_b[j] : plan->scratch[j];
      a.tensor[p.output_tensor[j]] = dst[j];
    }
    for (int d = 0; d < SODA_HIP_MAX_DIMS; ++d) a.dims[d] = d < p.dim ? dims[d] : 1;
    for (int s = 0; s < p.n_stages; ++s) {
      const Box& b = plan->boxes[it][p.n_inputs + s];
      for (int d = 0; d < SODA_HIP_MAX_DIMS; ++d) {
        a.box_lo[d] = d < p.dim ? vlo[d] - b.lo[d] : 0;
        a.box_hi[d] = d < p.dim ? dims[d] - vhi[d] - b.hi[d] : 1;
      }
      // everything the stage reads must be inside the array
      for (int w = 0; w < p.n_windows; ++w) {
        if (p.window[w].stage != p.n_inputs + s) continue;
        int rc = check_box_inside(plan, a, p.window[w].lo, p.window[w].hi, true);
        if (rc) return rc;
      }
      Launch l;
      bool empty;
      int rc = make_launch(plan, stage_kernel[s], a, &l, &empty);
      if (rc) return rc;
      if (!empty) list->push_back(l);
    }
    if (p.n_inputs == p.n_outputs) src = dst;
  }
  *max_depth_used = 1;
  return 0;
}

int launch_one(const soda_hip_plan* plan, const Launch& l, hipStream_t stream) {
  const soda_hip_kernel& desc = plan->kernels[l.kernel];
  soda_hip_args args = l.args;
  size_t size = sizeof args;
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args,
                    HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  HIP_TRY(SODA_HIP_ERR_DEVICE_RUN,
          hipModuleLaunchKernel(plan->funcs[l.kernel], l.grid[0], l.grid[1],
                                l.grid[2], desc.block[0], desc.block[1],
                                desc.block[2], l.lds_bytes, stream, nullptr, config));
  return 0;
}

// Streaming launches of the schedule (kernels that name a measured chunk, on boxes beyond
// the Infinity Cache): the calibrated (chunk, workgroups per CU) against the chunk's two
// neighbours on the calibration ladder and the other cap, each as a WHOLE sweep on this
// device; the fastest is kept per (kernel, box extents) when it beats the calibrated pair
// by more than 1 % (tools/calibrate.py measured those on one box; boxes differ).
template <typename TimeSweep>
int tune_streaming(soda_hip_plan* plan, const int64_t* dims, int iterate,
                   const int32_t* valid_lo, const int32_t* valid_hi, TimeSweep time_sweep) {
  if (plan->chunk_rows_override != 0 || plan->wgs_per_cu_cap != 0) return 0;
  std::vector<Launch> list;
  int depth = 0;
  int rc = build_schedule(plan, nullptr, nullptr, dims, iterate, valid_lo, valid_hi, &list,
                          &depth, nullptr, true);
  if (rc) return rc;
  static const int ladder[] = {8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256};
  const int n_ladder = (int)(sizeof ladder / sizeof ladder[0]);
  std::vector<std::array<int64_t, 5>> seen;
  for (const Launch& l : list) {
    const soda_hip_kernel& desc = plan->kernels[l.kernel];
    if (desc.fill_rows <= 0 || desc.stream_chunk <= 0 ||
        footprint_of(plan, l.args) <= kBeyondCacheBytes)
      continue;
    const std::array<int64_t, 5> key = stream_key(plan, l.kernel, l.args);
    if (std::find(seen.begin(), seen.end(), key) != seen.end()) continue;
    seen.push_back(key);
    plan->tuned_stream.erase(key);
    int at = 0;
    for (int i = 0; i < n_ladder; ++i)
      if (std::abs(ladder[i] - (int)desc.stream_chunk) < std::abs(ladder[at] - (int)desc.stream_chunk))
        at = i;
    const int cap0 = std::max(0, (int)desc.stream_wgs_per_cu);
    std::vector<std::array<int, 2>> candidates = {{(int)desc.stream_chunk, cap0}};
    for (int cap : {cap0, cap0 == 2 ? 0 : 2})
      for (int i : {at - 1, at, at + 1}) {
        if (i < 0 || i >= n_ladder) continue;
        const std::array<int, 2> c = {ladder[i], cap};
        if (std::find(candidates.begin(), candidates.end(), c) == candidates.end())
          candidates.push_back(c);
      }
    float incumbent = 0, best_ms = 0;
    size_t best = 0;
    for (size_t c = 0; c < candidates.size() && !rc; ++c) {
      if (c) plan->tuned_stream[key] = candidates[c];
      float ms = 0;
      rc = time_sweep(3, &ms);
      if (c == 0) incumbent = best_ms = ms;
      else if (ms < best_ms) { best_ms = ms; best = c; }
      if (tuning_env("SODA_HIP_DEBUG"))
        fprintf(stderr, "soda_hip: tune stream %s box %lld x %lld: chunk %d cap %d -> %.1f us\n",
                desc.name, (long long)key[1], (long long)key[2], candidates[c][0],
                candidates[c][1], ms * 1000.0);
    }
    if (!rc && best > 0 && best_ms < incumbent * 0.99f) plan->tuned_stream[key] = candidates[best];
    else plan->tuned_stream.erase(key);
  }
  return rc;
}

}  // namespace

extern "C" {

const char* soda_hip_error_name(int code) {
  switch (code) {
    case SODA_HIP_OK: return "ok";
    case SODA_HIP_ERR_GENERIC: return "generic_error";
    case SODA_HIP_ERR_BAD_ELEM_SIZE: return "bad_elem_size";
    case SODA_HIP_ERR_OUT_OF_BOUNDS: return "access_out_of_bounds";
    case SODA_HIP_ERR_EXTENTS_TOO_LARGE: return "buffer_extents_too_large";
    case SODA_HIP_ERR_CONSTRAINT: return "constraint_violated";
    case SODA_HIP_ERR_OUT_OF_MEMORY: return "out_of_memory";
    case SODA_HIP_ERR_NULL_ARGUMENT: return "buffer_argument_is_null";
    case SODA_HIP_ERR_COPY_TO_HOST: return "copy_to_host_failed";
    case SODA_HIP_ERR_COPY_TO_DEVICE: return "copy_to_device_failed";
    case SODA_HIP_ERR_DEVICE_MALLOC: return "device_malloc_failed";
    case SODA_HIP_ERR_DEVICE_SYNC: return "device_sync_failed";
    case SODA_HIP_ERR_DEVICE_FREE: return "device_free_failed";
    case SODA_HIP_ERR_NO_DEVICE: return "no_device_interface";
    case SODA_HIP_ERR_INTERNAL: return "internal_error";
    case SODA_HIP_ERR_DEVICE_RUN: return "device_run_failed";
    case SODA_HIP_ERR_COMPILE: return "kernel_compile_failed";
    case SODA_HIP_ERR_MODULE: return "bad_code_object";
    case SODA_HIP_ERR_NO_KERNEL: return "kernel_not_found";
    case SODA_HIP_ERR_MISMATCH: return "blob_program_mismatch";
    default: return "unknown_error";
  }
}

const char* soda_hip_last_error(void) { return g_last_error.c_str(); }
int soda_hip_abi_version(void) { return SODA_HIP_ABI_VERSION; }

int soda_hip_device_count(int* count) {
  if (!count) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(SODA_HIP_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return 0;
}

int soda_hip_set_device(int ordinal) {
  HIP_TRY(SODA_HIP_ERR_NO_DEVICE, hipSetDevice(ordinal));
  return 0;
}

int soda_hip_device_info(int ordinal, char* name, size_t name_cap, int* cu,
                         uint64_t* total_mem_bytes) {
  hipDeviceProp_t prop;
  HIP_TRY(SODA_HIP_ERR_NO_DEVICE, hipGetDeviceProperties(&prop, ordinal));
  if (name && name_cap) snprintf(name, name_cap, "%s", prop.gcnArchName);
  if (cu) *cu = prop.multiProcessorCount;
  if (total_mem_bytes) *total_mem_bytes = prop.totalGlobalMem;
  return 0;
}

int soda_hip_malloc(void** dev, size_t bytes) {
  if (!dev) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "dev is NULL");
  HIP_TRY(SODA_HIP_ERR_DEVICE_MALLOC, hipMalloc(dev, bytes ? bytes : 1));
  return 0;
}

int soda_hip_free(void* dev) {
  if (dev) HIP_TRY(SODA_HIP_ERR_DEVICE_FREE, hipFree(dev));
  return 0;
}

int soda_hip_memset(void* dev, int value, size_t bytes, void* stream) {
  HIP_TRY(SODA_HIP_ERR_DEVICE_RUN, hipMemsetAsync(dev, value, bytes, as_stream(stream)));
  return 0;
}

int soda_hip_memcpy_h2d(void* dev, const void* host, size_t bytes, void* stream) {
  HIP_TRY(SODA_HIP_ERR_COPY_TO_DEVICE,
          hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
  return 0;
}

int soda_hip_memcpy_d2h(void* host, const void* dev, size_t bytes, void* stream) {
  HIP_TRY(SODA_HIP_ERR_COPY_TO_HOST,
          hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  return 0;
}

int soda_hip_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
  HIP_TRY(SODA_HIP_ERR_DEVICE_RUN,
          hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
  return 0;
}

int soda_hip_stream_synchronize(void* stream) {
  HIP_TRY(SODA_HIP_ERR_DEVICE_SYNC, hipStreamSynchronize(as_stream(stream)));
  return 0;
}

// ---------------------------------------------------------------- modules
static int finish_module(soda_hip_module* m, soda_hip_module** out) {
  hipError_t e = hipModuleLoadData(&m->mod, m->image.data());
  if (e != hipSuccess) {
    delete m;
    return fail(SODA_HIP_ERR_MODULE, "hipModuleLoadData: %s", hipGetErrorString(e));
  }
  hipDeviceptr_t ptr = nullptr;
  size_t bytes = 0;
  if (hipModuleGetGlobal(&ptr, &bytes, m->mod, "soda_hip_meta") == hipSuccess && bytes) {
    m->meta.resize(bytes);
    if (hipMemcpyDtoH(&m->meta[0], ptr, bytes) != hipSuccess) m->meta.clear();
    const size_t z = m->meta.find('\0');
    if (z != std::string::npos) m->meta.resize(z);
  }
  (void)hipGetLastError();
  *out = m;
  return 0;
}

int soda_hip_module_load_data(const void* image, size_t bytes, soda_hip_module** module) {
  if (!image || !module) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  soda_hip_module* m = new soda_hip_module;
  m->image.assign((const char*)image, (const char*)image + bytes);
  return finish_module(m, module);
}

int soda_hip_module_load_file(const char* path, soda_hip_module** module) {
  if (!path || !module) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(SODA_HIP_ERR_MODULE, "cannot open %s", path);
  std::vector<char> data;
  char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
  fclose(f);
  if (data.empty()) return fail(SODA_HIP_ERR_MODULE, "%s is empty", path);
  return soda_hip_module_load_data(data.data(), data.size(), module);
}

int soda_hip_module_compile(const char* source, const char* arch,
                            const char* const* options, int n_options,
                            soda_hip_module** module) {
  if (!source || !module) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  std::string arch_flag = "--offload-arch=";
  if (arch && *arch) {
    arch_flag += arch;
  } else {
    int dev = 0;
    hipDeviceProp_t prop;
    HIP_TRY(SODA_HIP_ERR_NO_DEVICE, hipGetDevice(&dev));
    HIP_TRY(SODA_HIP_ERR_NO_DEVICE, hipGetDeviceProperties(&prop, dev));
    arch_flag += prop.gcnArchName;
  }
  std::vector<const char*> opts = {arch_flag.c_str(), "-O3", "-ffp-contract=off",
                                   "-std=c++17"};
  for (int i = 0; i < n_options; ++i) opts.push_back(options[i]);
  hiprtcProgram prog;
  hiprtcResult r = hiprtcCreateProgram(&prog, source, "soda_kernel.hip", 0, nullptr, nullptr);
  if (r != HIPRTC_SUCCESS)
    return fail(SODA_HIP_ERR_COMPILE, "hiprtcCreateProgram: %s", hiprtcGetErrorString(r));
  r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  if (r != HIPRTC_SUCCESS) {
    size_t log_size = 0;
    hiprtcGetProgramLogSize(prog, &log_size);
    std::string log(log_size, '\0');
    if (log_size) hiprtcGetProgramLog(prog, &log[0]);
    hiprtcDestroyProgram(&prog);
    return fail(SODA_HIP_ERR_COMPILE, "hiprtc: %s\n%.900s", hiprtcGetErrorString(r),
                log.c_str());
  }
  size_t code_size = 0;
  hiprtcGetCodeSize(prog, &code_size);
  soda_hip_module* m = new soda_hip_module;
  m->image.resize(code_size);
  hiprtcGetCode(prog, m->image.data());
  hiprtcDestroyProgram(&prog);
  return finish_module(m, module);
}

int soda_hip_module_image(const soda_hip_module* module, const void** image, size_t* bytes) {
  if (!module || !image || !bytes) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  *image = module->image.data();
  *bytes = module->image.size();
  return 0;
}

int soda_hip_module_meta(const soda_hip_module* module, char* buf, size_t cap, size_t* length) {
  if (!module) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "module is NULL");
  if (length) *length = module->meta.size();
  if (buf && cap) {
    const size_t n = std::min(cap - 1, module->meta.size());
    memcpy(buf, module->meta.data(), n);
    buf[n] = '\0';
  }
  return 0;
}

int soda_hip_module_unload(soda_hip_module* module) {
  if (!module) return 0;
  if (module->mod) HIP_TRY(SODA_HIP_ERR_MODULE, hipModuleUnload(module->mod));
  delete module;
  return 0;
}

// ------------------------------------------------------------------ plans
int soda_hip_plan_create(soda_hip_module* module, const soda_hip_program* program,
                         const soda_hip_kernel* kernels, int n_kernels,
                         soda_hip_plan** plan) {
  if (!module || !program || !kernels || !plan)
    return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  const soda_hip_program& p = *program;
  if (p.dim < 1 || p.dim > SODA_HIP_MAX_DIMS)
    return fail(SODA_HIP_ERR_CONSTRAINT, "dim %d not supported (1..%d)", p.dim,
                SODA_HIP_MAX_DIMS);
  if (p.n_inputs < 1 || p.n_stages < 1 || p.n_outputs < 1 ||
      p.n_inputs + p.n_stages > SODA_HIP_MAX_TENSORS || p.n_outputs > SODA_HIP_MAX_IO ||
      p.n_inputs > SODA_HIP_MAX_IO || p.n_windows < 1 || p.n_windows > SODA_HIP_MAX_WINDOWS)
    return fail(SODA_HIP_ERR_CONSTRAINT, "program descriptor out of range");
  if (n_kernels < 1 || n_kernels > SODA_HIP_MAX_KERNELS)
    return fail(SODA_HIP_ERR_CONSTRAINT, "n_kernels %d out of range", n_kernels);
  for (int j = 0; j < p.n_outputs; ++j)
    if (p.output_tensor[j] < p.n_inputs || p.output_tensor[j] >= p.n_inputs + p.n_stages)
      return fail(SODA_HIP_ERR_CONSTRAINT, "output %d is not a stage", j);
  for (int w = 0; w < p.n_windows; ++w) {
    const soda_hip_window& win = p.window[w];
    if (win.stage < p.n_inputs || win.stage >= p.n_inputs + p.n_stages ||
        win.parent < 0 || win.parent >= win.stage)
      return fail(SODA_HIP_ERR_CONSTRAINT, "window %d breaks execution order", w);
  }
  soda_hip_plan* pl = new soda_hip_plan;
  pl->module = module;
  pl->prog = p;
  pl->kernels.assign(kernels, kernels + n_kernels);
  pl->funcs.resize(n_kernels);
  for (int k = 0; k < n_kernels; ++k) {
    pl->kernels[k].name[sizeof(pl->kernels[k].name) - 1] = '\0';
    hipError_t e = hipModuleGetFunction(&pl->funcs[k], module->mod, pl->kernels[k].name);
    if (e != hipSuccess) {
      std::string name = pl->kernels[k].name;
      delete pl;
      return fail(SODA_HIP_ERR_NO_KERNEL, "kernel `%s` is not in the blob: %s",
                  name.c_str(), hipGetErrorString(e));
    }
    {
      // occupancy of this kernel on this device (MI355X_MICROARCH.md, register
      // files: 512 VGPRs per lane per SIMD, granule 8, at most 8 waves per SIMD)
      int regs = 0, dev = 0, cus = 256;
      (void)hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, pl->funcs[k]);
      hipDeviceProp_t prop;
      memset(&prop, 0, sizeof prop);
      if (hipGetDevice(&dev) == hipSuccess &&
          hipGetDeviceProperties(&prop, dev) == hipSuccess)
        cus = prop.multiProcessorCount;
      const int alloc = std::max(8, (regs + 7) / 8 * 8);
      const int waves_per_simd = std::max(1, std::min(8, 512 / alloc));
      const int threads = pl->kernels[k].block[0] * pl->kernels[k].block[1] *
                          pl->kernels[k].block[2];
      const int waves_per_block = std::max(1, (threads + 63) / 64);
      int per_cu = 4 * waves_per_simd / waves_per_block;
      // the runtime's own answer also knows the kernel's LDS and SGPR use
      int api = 0;
      if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(
              &api, pl->funcs[k], threads, 0) == hipSuccess && api > 0)
        per_cu = api;
      if (tuning_env("SODA_HIP_DEBUG"))
        fprintf(stderr, "soda_hip: kernel %s: %d VGPRs, %d workgroup(s) of %d "
                "wavefronts per CU\n", pl->kernels[k].name, regs, per_cu,
                waves_per_block);
      pl->resident_blocks.push_back(std::max(1, cus * per_cu));
      pl->cus = cus;
      int lds = 0;
      (void)hipFuncGetAttribute(&lds, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, pl->funcs[k]);
      pl->static_lds.push_back(std::max(0, lds));
      // (gfx950 has 160 KiB per CU; a runtime that reports the per-workgroup limit of
      // older parts here must not loosen the cap)
      pl->lds_per_cu = std::max<int64_t>(160 * 1024,
                                         (int64_t)prop.maxSharedMemoryPerMultiProcessor);
      if (tuning_env("SODA_HIP_DEBUG"))
        fprintf(stderr, "soda_hip: kernel %s: %d bytes of static LDS; device reports %lld "
                "bytes of LDS per CU\n", pl->kernels[k].name, lds,
                (long long)prop.maxSharedMemoryPerMultiProcessor);
    }
    const soda_hip_kernel& d = pl->kernels[k];
    if (d.block[0] < 1 || d.block[1] < 1 || d.block[2] < 1 ||
        (int64_t)d.block[0] * d.block[1] * d.block[2] > 1024) {
      delete pl;
      return fail(SODA_HIP_ERR_CONSTRAINT, "kernel %d has a bad block shape", k);
    }
  }
  if (const char* env = tuning_env("SODA_HIP_CHUNK_ROWS")) pl->chunk_rows_override = atoi(env);
  if (const char* env = tuning_env("SODA_HIP_WGS_PER_CU")) pl->wgs_per_cu_cap = atoi(env);
  if (const char* env = tuning_env("SODA_HIP_CHUNK_MIN"))
    pl->chunk_rows_min = std::max(4, atoi(env));
  *plan = pl;
  return 0;
}

int soda_hip_plan_destroy(soda_hip_plan* plan) {
  if (!plan) return 0;
  int rc = 0;
  for (void* ptr : plan->scratch)
    if (ptr && hipFree(ptr) != hipSuccess)
      rc = fail(SODA_HIP_ERR_DEVICE_FREE, "hipFree of plan scratch failed");
  for (void* ptr : plan->scratch_b)
    if (ptr && hipFree(ptr) != hipSuccess)
      rc = fail(SODA_HIP_ERR_DEVICE_FREE, "hipFree of plan scratch failed");
  if (plan->probe_buf) (void)hipFree(plan->probe_buf);
  if (plan->probe_t0) (void)hipEventDestroy(plan->probe_t0);
  if (plan->probe_t1) (void)hipEventDestroy(plan->probe_t1);
  if (plan->ev_main) (void)hipEventDestroy(plan->ev_main);
  if (plan->ev_landed) (void)hipEventDestroy(plan->ev_landed);
  if (plan->side) (void)hipStreamDestroy(plan->side);
  delete plan;
  return rc;
}

int soda_hip_plan_margins(const soda_hip_plan* plan, int iterations,
                          int32_t lo[SODA_HIP_MAX_DIMS], int32_t hi[SODA_HIP_MAX_DIMS]) {
  if (!plan || !lo || !hi) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  if (iterations < 0) return fail(SODA_HIP_ERR_CONSTRAINT, "iterations < 0");
  output_margins(const_cast<soda_hip_plan*>(plan), iterations, lo, hi);
  return 0;
}

int soda_hip_plan_set_max_depth(soda_hip_plan* plan, int max_depth) {
  if (!plan) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "plan is NULL");
  plan->max_depth = max_depth;
  return 0;
}

int soda_hip_plan_tune(soda_hip_plan* plan, void* const* in, void* const* out,
                       const int64_t dims[SODA_HIP_MAX_DIMS], int iterate,
                       const int32_t* valid_lo, const int32_t* valid_hi, void* stream) {
  if (!plan || !in || !out || !dims) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  const soda_hip_program& p = plan->prog;
  hipStream_t s = as_stream(stream);
  std::array<int64_t, 5> key;
  for (int d = 0; d < 4; ++d) key[d] = d < p.dim ? dims[d] : 1;
  key[4] = iterate;
  // candidate splits: the scheduler's own, and its answer when every deep kernel in
  // turn is made 12 % cheaper or dearer (the model ranks depths within a few percent
  // of each other; what really runs fastest depends on the device and the grid)
  std::vector<int> depths;
  for (const soda_hip_kernel& kd : plan->kernels)
    if (kd.kind == SODA_HIP_KERNEL_FUSED && kd.depth >= 4 && kd.depth <= iterate &&
        std::find(depths.begin(), depths.end(), kd.depth) == depths.end())
      depths.push_back(kd.depth);
  std::vector<std::vector<int>> candidates;
  plan->tuning = true;
  int rc = 0;
  for (int i = -1; i < 2 * (int)depths.size() && !rc; ++i) {
    plan->bias_depth = i < 0 ? 0 : depths[i / 2];
    plan->bias = i % 2 == 0 ? 0.88 : 1.12;
    std::vector<Launch> list;
    int depth = 0;
    rc = build_schedule(plan, nullptr, nullptr, dims, iterate, valid_lo, valid_hi, &list,
                        &depth, nullptr, true);
    std::vector<int> split;
    for (const Launch& l : list)
      if (plan->kernels[l.kernel].kind == SODA_HIP_KERNEL_FUSED)
        split.push_back(plan->kernels[l.kernel].depth);
    int total = 0;
    for (int d : split) total += d;
    if (!rc && total == iterate &&
        std::find(candidates.begin(), candidates.end(), split) == candidates.end())
      candidates.push_back(split);
  }
  plan->bias_depth = 0;
  plan->bias = 1.0;
  plan->tuning = false;
  if (rc) return rc;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
    return fail(SODA_HIP_ERR_DEVICE_RUN, "hipEventCreate failed");
  // the fastest of `runs` whole sweeps after one untimed one, in milliseconds
  auto time_sweep = [&](int runs, float* fastest) -> int {
    int e = 0;
    for (int run = 0; run <= runs && !e; ++run) {
      if (hipEventRecord(e0, s) != hipSuccess) e = fail(SODA_HIP_ERR_DEVICE_RUN, "hipEventRecord failed");
      if (!e) e = soda_hip_sweep(plan, in, out, dims, iterate, valid_lo, valid_hi, stream);
      if (!e && (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess))
        e = fail(SODA_HIP_ERR_DEVICE_SYNC, "timing a tuning sweep failed");
      float ms = 0;
      if (!e) (void)hipEventElapsedTime(&ms, e0, e1);
      if (run == 1 || (run > 1 && ms < *fastest)) *fastest = ms;
    }
    return e;
  };
  if (candidates.size() < 2) {      // one split only: the streaming launches remain
    rc = tune_streaming(plan, dims, iterate, valid_lo, valid_hi, time_sweep);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
  }
  // every candidate as a whole sweep, in context: one untimed run, then the faster
  // of two timed ones
  size_t best = 0;
  float best_ms = 0;
  for (size_t c = 0; c < candidates.size() && !rc; ++c) {
    plan->tuned_split[key] = candidates[c];
    float fastest = 0;
    rc = time_sweep(2, &fastest);
    if (tuning_env("SODA_HIP_DEBUG")) {
      fprintf(stderr, "soda_hip: tune %d iteration(s):", iterate);
      for (int d : candidates[c]) fprintf(stderr, " %d", d);
      fprintf(stderr, "  -> %.1f us\n", fastest * 1000.0);
    }
    if (c == 0 || fastest < best_ms) { best = c; best_ms = fastest; }
  }
  if (rc) plan->tuned_split.erase(key);
  else plan->tuned_split[key] = candidates[best];
  if (!rc) rc = tune_streaming(plan, dims, iterate, valid_lo, valid_hi, time_sweep);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

int soda_hip_plan_set_split(soda_hip_plan* plan, const int64_t dims[SODA_HIP_MAX_DIMS],
                            int iterate, const int32_t* depths, int n_depths) {
  if (!plan || !dims) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  const soda_hip_program& p = plan->prog;
  std::array<int64_t, 5> key;
  for (int d = 0; d < 4; ++d) key[d] = d < p.dim ? dims[d] : 1;
  key[4] = iterate;
  if (!depths || n_depths <= 0) {      // back to the scheduler's own choice
    plan->tuned_split.erase(key);
    return 0;
  }
  int total = 0;
  for (int i = 0; i < n_depths; ++i) {
    bool known = false;
    for (const soda_hip_kernel& kd : plan->kernels)
      known = known || (kd.kind == SODA_HIP_KERNEL_FUSED && kd.depth == depths[i]);
    if (!known)
      return fail(SODA_HIP_ERR_NO_KERNEL, "no fused kernel of depth %d in the blob",
                  (int)depths[i]);
    total += depths[i];
  }
  if (total != iterate)
    return fail(SODA_HIP_ERR_CONSTRAINT, "the depths add up to %d, not to iterate = %d",
                total, iterate);
  plan->tuned_split[key] = std::vector<int>(depths, depths + n_depths);
  return 0;
}

int soda_hip_plan_set_out_final_only(soda_hip_plan* plan, int on) {
  if (!plan) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "plan is NULL");
  plan->out_final_only = on != 0;
  return 0;
}

int soda_hip_plan_schedule(soda_hip_plan* plan, const int64_t dims[SODA_HIP_MAX_DIMS],
                           int iterate, const int32_t* valid_lo,
                           const int32_t* valid_hi, int32_t* kernel_index,
                           double* est_us, int capacity, int* n_launches) {
  if (!plan || !dims || !n_launches) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  std::vector<Launch> list;
  int depth = 0;
  int rc = build_schedule(plan, nullptr, nullptr, dims, iterate, valid_lo, valid_hi, &list,
                          &depth, nullptr, true);
  if (rc) return rc;
  *n_launches = (int)list.size();
  for (int i = 0; i < (int)list.size() && i < capacity; ++i) {
    if (kernel_index) kernel_index[i] = list[i].kernel;
    if (est_us) est_us[i] = list[i].est_us;
  }
  return 0;
}

int soda_hip_sweep(soda_hip_plan* plan, void* const* in, void* const* out,
                   const int64_t dims[SODA_HIP_MAX_DIMS], int iterate,
                   const int32_t* valid_lo, const int32_t* valid_hi, void* stream) {
  if (!plan || !in || !out || !dims) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  std::vector<Launch> list;
  int depth = 0;
  int rc = build_schedule(plan, in, out, dims, iterate, valid_lo, valid_hi, &list, &depth,
                          as_stream(stream));
  if (rc) return rc;
  for (const Launch& l : list) {
    rc = launch_one(plan, l, as_stream(stream));
    if (rc) return rc;
  }
  return 0;
}

int soda_hip_sweep_timed(soda_hip_plan* plan, void* const* in, void* const* out,
                         const int64_t dims[SODA_HIP_MAX_DIMS], int iterate,
                         int warmup, int repeats, void* stream,
                         soda_hip_timing* timing) {
  if (!plan || !in || !out || !dims) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  if (repeats < 1) return fail(SODA_HIP_ERR_CONSTRAINT, "repeats must be >= 1");
  hipStream_t s = as_stream(stream);
  std::vector<Launch> list;
  int depth = 0;
  int rc = build_schedule(plan, in, out, dims, iterate, nullptr, nullptr, &list, &depth, s);
  if (rc) return rc;
  for (int w = 0; w < warmup; ++w)
    for (const Launch& l : list)
      if ((rc = launch_one(plan, l, s))) return rc;
  // one event before every launch and one after the last, per repeat
  const size_t per = list.size() + 1;
  std::vector<hipEvent_t> ev(per * repeats, nullptr);
  for (auto& e : ev)
    if (!rc && hipEventCreate(&e) != hipSuccess) {
      e = nullptr;
      rc = fail(SODA_HIP_ERR_DEVICE_RUN, "hipEventCreate failed");
    }
  for (int r = 0; r < repeats && !rc; ++r) {
    for (size_t i = 0; i < list.size() && !rc; ++i) {
      if (hipEventRecord(ev[r * per + i], s) != hipSuccess)
        rc = fail(SODA_HIP_ERR_DEVICE_RUN, "hipEventRecord failed");
      else
        rc = launch_one(plan, list[i], s);
    }
    if (!rc && hipEventRecord(ev[r * per + list.size()], s) != hipSuccess)
      rc = fail(SODA_HIP_ERR_DEVICE_RUN, "hipEventRecord failed");
  }
  if (!rc && hipStreamSynchronize(s) != hipSuccess)
    rc = fail(SODA_HIP_ERR_DEVICE_SYNC, "hipStreamSynchronize failed: %s",
              hipGetErrorString(hipGetLastError()));
  if (!rc && timing) {
    memset(timing, 0, sizeof *timing);
    double total_ms = 0;
    // every launch at its FASTEST repeat (events between launches add gaps and a
    // first repeat runs at other clocks: a mean over-states the kernels)
    std::vector<float> fastest(list.size(), 0.f);
    std::map<int, std::pair<double, int>> per_kernel;
    for (int r = 0; r < repeats; ++r) {
      float ms = 0;
      (void)hipEventElapsedTime(&ms, ev[r * per], ev[r * per + list.size()]);
      total_ms += ms;
      for (size_t i = 0; i < list.size(); ++i) {
        float k_ms = 0;
        (void)hipEventElapsedTime(&k_ms, ev[r * per + i], ev[r * per + i + 1]);
        if (r == 0 || k_ms < fastest[i]) fastest[i] = k_ms;
      }
    }
    double fastest_ms = 0;
    for (size_t i = 0; i < list.size(); ++i) {
      auto& slot = per_kernel[list[i].kernel];
      slot.first += fastest[i];
      slot.second += 1;
      fastest_ms += fastest[i];
    }
    if (tuning_env("SODA_HIP_LAUNCH_TRACE"))   // tools/: every launch, fastest repeat
      for (size_t i = 0; i < list.size(); ++i) {
        const soda_hip_args& a = list[i].args;
        fprintf(stderr, "soda_hip: launch %3zu %-28s %8.1f us (model %7.1f)  box %lld x %lld x %lld  "
                "grid %u x %u x %u  chunk %lld  fill %d  resident %d  lds %u  rounds %lld\n", i,
                plan->kernels[list[i].kernel].name,
                fastest[i] * 1000.0, list[i].est_us, (long long)(a.box_hi[0] - a.box_lo[0]),
                (long long)(a.box_hi[1] - a.box_lo[1]), (long long)(a.box_hi[2] - a.box_lo[2]),
                list[i].grid[0], list[i].grid[1], list[i].grid[2], (long long)a.param[0],
                plan->kernels[list[i].kernel].fill_rows,
                list[i].resident > 0 ? (int)list[i].resident
                                     : plan->resident_blocks[list[i].kernel],
                list[i].lds_bytes, list[i].rounds);
      }
    timing->kernel_us = total_ms * 1000.0 / repeats;
    timing->fastest_us = fastest_ms * 1000.0;
    timing->launches = (int)list.size();
    timing->max_depth = depth;
    int best = -1;
    for (auto& kv : per_kernel)
      if (best < 0 || kv.second.first > per_kernel[best].first) best = kv.first;
    if (best >= 0) {
      timing->dominant_us = per_kernel[best].first * 1000.0;
      timing->dominant_launches = per_kernel[best].second;
      snprintf(timing->dominant_name, sizeof timing->dominant_name, "%s",
               plan->kernels[best].name);
    }
  }
  for (auto& e : ev)
    if (e) (void)hipEventDestroy(e);
  return rc;
}

// ------------------------------------------------------------- clock probe
int soda_hip_clock_probe_start(soda_hip_plan* plan, int spins) {
  if (!plan) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "plan is NULL");
  if (spins < 1) return fail(SODA_HIP_ERR_CONSTRAINT, "spins must be >= 1");
  if (plan->probe_running) return fail(SODA_HIP_ERR_CONSTRAINT, "a clock probe is running");
  if (!plan->probe &&
      hipModuleGetFunction(&plan->probe, plan->module->mod, "soda_hip_clock_probe") != hipSuccess) {
    plan->probe = nullptr;
    (void)hipGetLastError();
    return fail(SODA_HIP_ERR_NO_KERNEL, "the blob holds no soda_hip_clock_probe (built "
                "before ABI 7)");
  }
  if (!plan->side &&
      hipStreamCreateWithFlags(&plan->side, hipStreamNonBlocking) != hipSuccess) {
    plan->side = nullptr;
    return fail(SODA_HIP_ERR_DEVICE_RUN, "hipStreamCreate failed");
  }
  if (!plan->probe_buf) HIP_TRY(SODA_HIP_ERR_DEVICE_MALLOC, hipMalloc(&plan->probe_buf, 16));
  HIP_TRY(SODA_HIP_ERR_DEVICE_RUN, hipMemsetAsync(plan->probe_buf, 0, 16, plan->side));
  struct { void* out; int spins; } args = {plan->probe_buf, spins};
  size_t size = sizeof args;
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE,
                    &size, HIP_LAUNCH_PARAM_END};
  // the wall time of the probe comes from events around it (the realtime counter it
  // reads is nominally 100 MHz; measured against events it is what calibrates it)
  if (!plan->probe_t0 && hipEventCreate(&plan->probe_t0) != hipSuccess) {
    plan->probe_t0 = nullptr;
    return fail(SODA_HIP_ERR_DEVICE_RUN, "hipEventCreate failed");
  }
  if (!plan->probe_t1 && hipEventCreate(&plan->probe_t1) != hipSuccess) {
    plan->probe_t1 = nullptr;
    return fail(SODA_HIP_ERR_DEVICE_RUN, "hipEventCreate failed");
  }
  HIP_TRY(SODA_HIP_ERR_DEVICE_RUN, hipEventRecord(plan->probe_t0, plan->side));
  HIP_TRY(SODA_HIP_ERR_DEVICE_RUN,
          hipModuleLaunchKernel(plan->probe, 1, 1, 1, 64, 1, 1, 0, plan->side, nullptr, config));
  HIP_TRY(SODA_HIP_ERR_DEVICE_RUN, hipEventRecord(plan->probe_t1, plan->side));
  plan->probe_running = true;
  return 0;
}

int soda_hip_clock_probe_finish(soda_hip_plan* plan, double* shader_ghz, double* seconds) {
  if (!plan || !shader_ghz) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  if (!plan->probe_running) return fail(SODA_HIP_ERR_CONSTRAINT, "no clock probe is running");
  plan->probe_running = false;
  HIP_TRY(SODA_HIP_ERR_DEVICE_SYNC, hipStreamSynchronize(plan->side));
  unsigned long long got[2] = {0, 0};
  HIP_TRY(SODA_HIP_ERR_COPY_TO_HOST, hipMemcpy(got, plan->probe_buf, 16, hipMemcpyDeviceToHost));
  if (!got[1]) return fail(SODA_HIP_ERR_DEVICE_RUN, "the clock probe reported no time");
  // s_memrealtime ticks at a nominal 100 MHz; the events around the probe give the wall
  // time independently (the kernel is one wavefront that starts at once on the non-blocking
  // stream, so both spans agree to a few microseconds when the counter's rate is as named)
  double elapsed = (double)got[1] / 100.0e6;
  float ms = 0;
  if (hipEventElapsedTime(&ms, plan->probe_t0, plan->probe_t1) == hipSuccess && ms > 0) {
    const double by_events = ms * 1e-3;
    if (tuning_env("SODA_HIP_DEBUG"))
      fprintf(stderr, "soda_hip: clock probe: %llu shader cycles, %llu realtime ticks = %.3f ms "
              "at 100 MHz, events %.3f ms\n", got[0], got[1], elapsed * 1e3, by_events * 1e3);
    // a probe that waited for a wave slot makes the event span LONGER than its own count;
    // take the counter unless the two disagree by more than the realtime clock could
    if (by_events < elapsed * 0.97) elapsed = by_events;
  }
  *shader_ghz = (double)got[0] / elapsed / 1e9;
  if (seconds) *seconds = elapsed;
  return 0;
}

// ------------------------------------------------------- multi-GPU slab driver
namespace {

// RCCL is resolved at first use: a single-GPU caller never loads it.
struct Rccl {
  int (*group_start)() = nullptr;
  int (*group_end)() = nullptr;
  int (*send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*error_string)(int) = nullptr;
  int (*comm_abort)(void*) = nullptr;
  bool ok = false;
};

const Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return x;
    x.group_start = (int (*)())dlsym(h, "ncclGroupStart");
    x.group_end = (int (*)())dlsym(h, "ncclGroupEnd");
    x.send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclSend");
    x.recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclRecv");
    x.error_string = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    x.comm_abort = (int (*)(void*))dlsym(h, "ncclCommAbort");
    x.ok = x.group_start && x.group_end && x.send && x.recv;
    return x;
  }();
  return r;
}

struct SlabGeometry {
  int64_t own, ghost_lo, ghost_hi, extent, row_bytes;
  bool has_lo, has_hi;
};

int slab_geometry(const soda_hip_plan* plan, const soda_hip_slab* s, SlabGeometry* g) {
  const soda_hip_program& p = plan->prog;
  if (p.n_inputs != 1 || p.n_outputs != 1)
    return fail(SODA_HIP_ERR_CONSTRAINT, "slabs: one-input one-output programs");
  if (s->world < 1 || s->rank < 0 || s->rank >= s->world || s->exchange < 1 ||
      s->reach_lo < 0 || s->reach_hi < 0)
    return fail(SODA_HIP_ERR_CONSTRAINT, "slab descriptor out of range");
  g->own = s->own_last - s->own_first;
  g->has_lo = s->rank > 0;
  g->has_hi = s->rank < s->world - 1;
  g->ghost_lo = g->has_lo ? (int64_t)s->exchange * s->reach_lo : 0;
  g->ghost_hi = g->has_hi ? (int64_t)s->exchange * s->reach_hi : 0;
  // a ghost region deeper than a neighbour's own rows would ship rows it does
  // not own (runtime/dist.py: SlabPlan raises for the same reason)
  if (g->own < 1 || (s->world > 1 && g->own < (int64_t)s->exchange *
                                                 std::max(s->reach_lo, s->reach_hi)))
    return fail(SODA_HIP_ERR_CONSTRAINT,
                "slab of %lld own rows is thinner than its ghost regions (%d x %d)",
                (long long)g->own, s->exchange, std::max(s->reach_lo, s->reach_hi));
  g->extent = g->ghost_lo + g->own + g->ghost_hi;
  g->row_bytes = p.elem_size[0];
  for (int d = 0; d < p.dim - 1; ++d) g->row_bytes *= s->dims[d];
  return 0;
}

// ---- slabs re-cut to the shrinking valid box (runtime/dist.py: RecutPlan) ----
struct Rows {
  int64_t lo = 0, hi = 0;
  bool empty() const { return hi <= lo; }
};

Rows intersect(const Rows& a, const Rows& b) {
  Rows r;
  r.lo = std::max(a.lo, b.lo);
  r.hi = std::min(a.hi, b.hi);
  return r;
}

// world + 1 cut points of [lo, hi): as even as possible, the longer shares first
std::vector<int64_t> even_cut(int64_t lo, int64_t hi, int world) {
  const int64_t extent = std::max<int64_t>(0, hi - lo);
  const int64_t base = extent / world, extra = extent % world;
  std::vector<int64_t> pts(world + 1, lo);
  for (int r = 0; r < world; ++r) pts[r + 1] = pts[r] + base + (r < extra ? 1 : 0);
  return pts;
}

struct RecutStep {
  int done = 0, step = 0;
  std::vector<Rows> owned;      // per rank: rows of the INPUT level it holds
  std::vector<int64_t> cuts;    // world + 1 cut points of the OUTPUT level's rows
  std::vector<Rows> need;       // per rank: rows of the input level it reads (empty: none)
};

struct RecutTable {
  std::vector<RecutStep> steps;
  std::vector<Rows> final;      // per rank: rows of the result
  int64_t base = 0, extent = 0; // this rank's arrays span global rows [base, base + extent)
  int64_t row_bytes = 0;
};

int recut_table(const soda_hip_plan* plan, const soda_hip_slab* s, int iterate, RecutTable* t) {
  const soda_hip_program& p = plan->prog;
  if (p.n_inputs != 1 || p.n_outputs != 1)
    return fail(SODA_HIP_ERR_CONSTRAINT, "slabs: one-input one-output programs");
  if (s->world < 1 || s->rank < 0 || s->rank >= s->world || s->exchange < 1 ||
      s->reach_lo < 0 || s->reach_hi < 0 || iterate < 1)
    return fail(SODA_HIP_ERR_CONSTRAINT, "slab descriptor out of range");
  const int64_t rows = s->dims[p.dim - 1];
  if (rows < 1) return fail(SODA_HIP_ERR_CONSTRAINT, "slab descriptor: %lld rows", (long long)rows);
  const std::vector<int64_t> level0 = even_cut(0, rows, s->world);
  if (s->own_first != level0[s->rank] || s->own_last != level0[s->rank + 1])
    return fail(SODA_HIP_ERR_CONSTRAINT,
                "re-cut slabs: rank %d of %d must be handed rows [%lld, %lld) of %lld (the even "
                "cut), not [%lld, %lld)", s->rank, s->world, (long long)level0[s->rank],
                (long long)level0[s->rank + 1], (long long)rows, (long long)s->own_first,
                (long long)s->own_last);
  std::vector<Rows> level(s->world);
  for (int r = 0; r < s->world; ++r) { level[r].lo = level0[r]; level[r].hi = level0[r + 1]; }
  t->steps.clear();
  int64_t lo_hull = s->own_first, hi_hull = s->own_last;
  for (int done = 0; done < iterate;) {
    RecutStep st;
    st.done = done;
    st.step = std::min(s->exchange, iterate - done);
    const int64_t lo = (int64_t)(done + st.step) * s->reach_lo;
    const int64_t hi = rows - (int64_t)(done + st.step) * s->reach_hi;
    st.cuts = even_cut(lo, std::max(lo, hi), s->world);
    st.owned = level;
    st.need.assign(s->world, Rows{});
    for (int r = 0; r < s->world; ++r) {
      if (st.cuts[r + 1] <= st.cuts[r]) continue;
      st.need[r].lo = st.cuts[r] - (int64_t)st.step * s->reach_lo;
      st.need[r].hi = st.cuts[r + 1] + (int64_t)st.step * s->reach_hi;
    }
    if (!st.need[s->rank].empty()) {
      lo_hull = std::min(lo_hull, st.need[s->rank].lo);
      hi_hull = std::max(hi_hull, st.need[s->rank].hi);
    }
    for (int r = 0; r < s->world; ++r) { level[r].lo = st.cuts[r]; level[r].hi = st.cuts[r + 1]; }
    done += st.step;
    t->steps.push_back(st);
  }
  t->final = level;
  t->base = lo_hull;
  t->extent = hi_hull - lo_hull;
  t->row_bytes = p.elem_size[0];
  for (int d = 0; d < p.dim - 1; ++d) t->row_bytes *= s->dims[d];
  return 0;
}

// Super-step i, bands first (RecutPlan.pieces): the rows other ranks read in super-step
// i + 1 come first, the interior afterwards.  false: nothing to gain (the last super-step,
// no output rows, nobody waiting, or bands that meet).
bool recut_pieces(const RecutTable& t, const soda_hip_slab* s, size_t i, std::vector<Rows>* bands,
                  Rows* interior) {
  if (i + 1 >= t.steps.size()) return false;
  const RecutStep& st = t.steps[i];
  const RecutStep& next = t.steps[i + 1];
  const int64_t lo = st.cuts[s->rank], hi = st.cuts[s->rank + 1];
  if (hi <= lo) return false;
  int64_t b_lo = lo, b_hi = hi;
  for (int q = 0; q < s->rank; ++q)
    if (!next.need[q].empty() && next.need[q].hi > lo) b_lo = std::max(b_lo, next.need[q].hi);
  for (int q = s->rank + 1; q < s->world; ++q)
    if (!next.need[q].empty() && next.need[q].lo < hi) b_hi = std::min(b_hi, next.need[q].lo);
  b_lo = std::min(b_lo, hi);
  b_hi = std::max(b_hi, lo);
  if ((b_lo == lo && b_hi == hi) || b_lo >= b_hi) return false;
  bands->clear();
  if (b_lo > lo) bands->push_back(Rows{lo, b_lo});
  if (b_hi < hi) bands->push_back(Rows{b_hi, hi});
  interior->lo = b_lo;
  interior->hi = b_hi;
  return true;
}

}  // namespace

int soda_hip_slab_exchange(int64_t rows, int world, int reach_lo, int reach_hi,
                           int wanted, int* exchange) {
  if (!exchange) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  if (rows < 1 || world < 1 || wanted < 1 || reach_lo < 0 || reach_hi < 0)
    return fail(SODA_HIP_ERR_CONSTRAINT, "slab figures out of range");
  const int64_t reach = std::max(1, std::max(reach_lo, reach_hi));
  const int64_t smallest = rows / world;
  if (world > 1 && smallest < reach)
    return fail(SODA_HIP_ERR_CONSTRAINT,
                "cannot cut %lld rows into %d slabs: the smallest slab (%lld rows) is "
                "thinner than the stencil reach (%lld)", (long long)rows, world,
                (long long)smallest, (long long)reach);
  *exchange = world > 1 ? (int)std::max<int64_t>(1, std::min<int64_t>(wanted, smallest / reach))
                        : wanted;
  return 0;
}

int soda_hip_slab_extent(const soda_hip_plan* plan, const soda_hip_slab* slab,
                         int64_t local_dims[SODA_HIP_MAX_DIMS], int64_t* ghost_lo,
                         int64_t* ghost_hi) {
  if (!plan || !slab || !local_dims) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  if (slab->cut != SODA_HIP_SLAB_CUT_STATIC)
    return fail(SODA_HIP_ERR_CONSTRAINT, "soda_hip_slab_extent describes the static cut; a "
                "re-cut run's arrays depend on the iteration count: soda_hip_slab_layout");
  SlabGeometry g;
  int rc = slab_geometry(plan, slab, &g);
  if (rc) return rc;
  for (int d = 0; d < SODA_HIP_MAX_DIMS; ++d)
    local_dims[d] = d < plan->prog.dim ? slab->dims[d] : 1;
  local_dims[plan->prog.dim - 1] = g.extent;
  if (ghost_lo) *ghost_lo = g.ghost_lo;
  if (ghost_hi) *ghost_hi = g.ghost_hi;
  return 0;
}

int soda_hip_slab_layout(const soda_hip_plan* plan, const soda_hip_slab* slab, int iterate,
                         int64_t local_dims[SODA_HIP_MAX_DIMS], int64_t* input_offset,
                         int64_t* result_first, int64_t* result_last,
                         int64_t* result_offset) {
  if (!plan || !slab || !local_dims) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  if (slab->cut != SODA_HIP_SLAB_CUT_STATIC && slab->cut != SODA_HIP_SLAB_CUT_RECUT)
    return fail(SODA_HIP_ERR_CONSTRAINT, "slab cut %d", (int)slab->cut);
  for (int d = 0; d < SODA_HIP_MAX_DIMS; ++d)
    local_dims[d] = d < plan->prog.dim ? slab->dims[d] : 1;
  if (slab->cut == SODA_HIP_SLAB_CUT_STATIC) {
    SlabGeometry g;
    int rc = slab_geometry(plan, slab, &g);
    if (rc) return rc;
    local_dims[plan->prog.dim - 1] = g.extent;
    if (input_offset) *input_offset = g.ghost_lo;
    if (result_first) *result_first = slab->own_first;
    if (result_last) *result_last = slab->own_last;
    if (result_offset) *result_offset = g.ghost_lo;
    return 0;
  }
  RecutTable t;
  int rc = recut_table(plan, slab, iterate, &t);
  if (rc) return rc;
  local_dims[plan->prog.dim - 1] = t.extent;
  if (input_offset) *input_offset = slab->own_first - t.base;
  if (result_first) *result_first = t.final[slab->rank].lo;
  if (result_last) *result_last = t.final[slab->rank].hi;
  if (result_offset) *result_offset = t.final[slab->rank].lo - t.base;
  return 0;
}

int soda_hip_run_slab(soda_hip_plan* plan, const soda_hip_slab* slab, void* comm,
                      void* a, void* b, void* c, int iterate, void* stream,
                      void** result, int* exchanges) {
  if (!plan || !slab || !a || !b || !c || !result)
    return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  if (slab->world > 1 && !comm)
    return fail(SODA_HIP_ERR_NULL_ARGUMENT, "world %d needs an RCCL communicator", slab->world);
  if (slab->world > 1 && !rccl().ok)
    return fail(SODA_HIP_ERR_NO_DEVICE, "librccl.so could not be loaded: %s", dlerror());
  // Everything that can be wrong with the call itself is found before the first message
  // is enqueued: such an error leaves the communicator alone (the peers have not been
  // promised anything yet - the caller's own rendezvous, or its next call, sees it).
  if (iterate < 1) return fail(SODA_HIP_ERR_CONSTRAINT, "iterate must be >= 1");
  if (slab->order != SODA_HIP_SLAB_SERIAL && slab->order != SODA_HIP_SLAB_BANDS_FIRST)
    return fail(SODA_HIP_ERR_CONSTRAINT, "slab order %d", (int)slab->order);
  if (slab->cut != SODA_HIP_SLAB_CUT_STATIC && slab->cut != SODA_HIP_SLAB_CUT_RECUT)
    return fail(SODA_HIP_ERR_CONSTRAINT, "slab cut %d", (int)slab->cut);
  SlabGeometry g{};
  RecutTable table;
  const bool recut = slab->cut == SODA_HIP_SLAB_CUT_RECUT;
  int rc = recut ? recut_table(plan, slab, iterate, &table) : slab_geometry(plan, slab, &g);
  if (rc) return rc;
  const bool overlapped = slab->order == SODA_HIP_SLAB_BANDS_FIRST && slab->world > 1;
  if (overlapped) {
    // the stream and the two events of the bands-first order, each under its own check (the
    // clock probe creates the same stream; a half-built set must be completed, not skipped)
    if (!plan->side && hipStreamCreateWithFlags(&plan->side, hipStreamNonBlocking) != hipSuccess) {
      plan->side = nullptr;
      return fail(SODA_HIP_ERR_DEVICE_RUN, "side stream for the exchange: %s",
                  hipGetErrorString(hipGetLastError()));
    }
    if (!plan->ev_main &&
        hipEventCreateWithFlags(&plan->ev_main, hipEventDisableTiming) != hipSuccess) {
      plan->ev_main = nullptr;
      return fail(SODA_HIP_ERR_DEVICE_RUN, "event for the exchange stream: %s",
                  hipGetErrorString(hipGetLastError()));
    }
    if (!plan->ev_landed &&
        hipEventCreateWithFlags(&plan->ev_landed, hipEventDisableTiming) != hipSuccess) {
      plan->ev_landed = nullptr;
      return fail(SODA_HIP_ERR_DEVICE_RUN, "event for the exchange stream: %s",
                  hipGetErrorString(hipGetLastError()));
    }
  }
  // From here on a failure of THIS rank may leave peers waiting in ncclRecv for rows it
  // will never send.  abort_on_error: after a failure of this rank's OWN (a launch, an
  // allocation - not an error RCCL reports, which may be somebody's abort of this very
  // communicator) the communicator is aborted before the error is returned (best effort -
  // ncclCommAbort is local to the rank, include/soda_hip.h); otherwise the communicator
  // is the caller's to abort, for every rank of its process.
  bool rccl_failed = false;     // the error came from RCCL itself (e.g. an aborted communicator)
  auto give_up = [&](int rc) {
    if (rc && !rccl_failed && slab->abort_on_error && slab->world > 1 && comm &&
        rccl().comm_abort) {
      const std::string keep = g_last_error;
      (void)rccl().comm_abort(comm);
      g_last_error = keep + " (communicator aborted)";
    }
    return rc;
  };
  const soda_hip_program& p = plan->prog;
  const int last = p.dim - 1;
  hipStream_t s = as_stream(stream);
  int64_t local_dims[SODA_HIP_MAX_DIMS] = {1, 1, 1, 1};
  for (int d = 0; d < p.dim; ++d) local_dims[d] = slab->dims[d];
  local_dims[last] = recut ? table.extent : g.extent;
  const int64_t row_bytes = recut ? table.row_bytes : g.row_bytes;
  const int64_t send_down = !recut && g.has_lo ? (int64_t)slab->exchange * slab->reach_hi : 0;
  const int64_t send_up = !recut && g.has_hi ? (int64_t)slab->exchange * slab->reach_lo : 0;
  // one message = rows [first, first + rows) of the LOCAL array, to or from a peer
  struct Message { bool send; int peer; int64_t first, rows; };
  auto static_messages = [&]() {
    std::vector<Message> m;
    const int64_t first_own = g.ghost_lo, last_own = g.ghost_lo + g.own;
    // lower neighbour: it needs our first rows, we need its last ones
    if (g.has_lo && send_down) m.push_back({true, slab->rank - 1, first_own, send_down});
    if (g.has_lo && g.ghost_lo) m.push_back({false, slab->rank - 1, 0, g.ghost_lo});
    if (g.has_hi && send_up) m.push_back({true, slab->rank + 1, last_own - send_up, send_up});
    if (g.has_hi && g.ghost_hi) m.push_back({false, slab->rank + 1, last_own, g.ghost_hi});
    return m;
  };
  // before super-step i of a re-cut run: to every rank the rows it reads and we hold, from
  // every rank the rows we read and it holds - ghost rows and rows changing owner alike
  // (both sides derive a pair's rows from the same table; ascending peers, sends first)
  auto recut_messages = [&](size_t i) {
    std::vector<Message> m;
    const RecutStep& st = table.steps[i];
    const Rows& mine = st.owned[slab->rank];
    for (int pass = 0; pass < 2; ++pass)
      for (int q = 0; q < slab->world; ++q) {
        if (q == slab->rank) continue;
        const Rows rows = pass == 0 ? intersect(st.need[q], mine)
                                    : intersect(st.need[slab->rank], st.owned[q]);
        const bool wanted = pass == 0 ? !st.need[q].empty() && !mine.empty()
                                      : !st.need[slab->rank].empty() && !st.owned[q].empty();
        if (wanted && !rows.empty())
          m.push_back({pass == 0, q, rows.lo - table.base, rows.hi - rows.lo});
      }
    return m;
  };
  auto exchange_rows = [&](char* array, const std::vector<Message>& messages,
                           hipStream_t on) -> int {
    if (slab->world == 1 || messages.empty()) return 0;
    const Rccl& r = rccl();
    int e = r.group_start();
    for (const Message& m : messages) {
      if (e) break;
      char* at = array + m.first * row_bytes;
      e = m.send ? r.send(at, (size_t)(m.rows * row_bytes), 0, m.peer, comm, on)
                 : r.recv(at, (size_t)(m.rows * row_bytes), 0, m.peer, comm, on);
    }
    const int e2 = r.group_end();
    if (e || e2) {
      rccl_failed = true;
      return fail(SODA_HIP_ERR_DEVICE_RUN, "RCCL ghost exchange failed: %s",
                  r.error_string ? r.error_string(e ? e : e2) : "?");
    }
    return 0;
  };
  // Bands-first order (runtime/dist.py: StreamSchedule; band_plan / RecutPlan.pieces):
  // every super-step but the last first sweeps the bands of rows other ranks are waiting
  // for, hands them to the exchange of the NEXT super-step on a stream the plan owns, and
  // sweeps the interior meanwhile.  A piece's intermediate launches must not write rows of
  // `dst` another piece has finished (they are being sent): pieces run with out_final_only.
  bool landed_pending = false;      // an exchange on the side stream main has not waited for
  auto exchange = [&](char* array, const std::vector<Message>& messages) -> int {
    if (!overlapped) return exchange_rows(array, messages, s);
    // the rows to be sent were produced on the main stream: the side stream follows
    // everything enqueued there so far
    if (hipEventRecord(plan->ev_main, s) != hipSuccess ||
        hipStreamWaitEvent(plan->side, plan->ev_main, 0) != hipSuccess)
      return fail(SODA_HIP_ERR_DEVICE_RUN, "ordering the exchange stream failed");
    int e = exchange_rows(array, messages, plan->side);
    if (e) return e;
    if (hipEventRecord(plan->ev_landed, plan->side) != hipSuccess)
      return fail(SODA_HIP_ERR_DEVICE_RUN, "hipEventRecord failed");
    landed_pending = true;
    return 0;
  };
  auto ghosts_have_landed = [&]() -> int {      // before a sweep reads ghost rows
    if (!landed_pending) return 0;
    landed_pending = false;
    if (hipStreamWaitEvent(s, plan->ev_landed, 0) != hipSuccess)
      return fail(SODA_HIP_ERR_DEVICE_RUN, "hipStreamWaitEvent failed");
    return 0;
  };
  const bool was_final_only = plan->out_final_only;
  // test hook (SODA_HIP_TUNING=1 only): rank R fails at its K-th super-step
  int fail_rank = -1, fail_at = -1;
  if (const char* env = tuning_env("SODA_HIP_FAIL_RANK")) fail_rank = atoi(env);
  if (const char* env = tuning_env("SODA_HIP_FAIL_SUPERSTEP")) fail_at = atoi(env);
  void* src = a;
  void* cycle[2] = {b, c};
  int done = 0, k = 0, count = 0;
  bool pending = false;            // src's ghost rows are (being) filled already
  // the sub-array of local rows [r0, r1) swept `step` iterations with the given outer
  // margins (0 = the side is cut inside valid rows)
  auto sweep_rows = [&](void* from, void* to, int64_t r0, int64_t r1, int step,
                        const int32_t* lo, const int32_t* hi, bool final_only) -> int {
    int64_t dims_piece[SODA_HIP_MAX_DIMS];
    for (int d = 0; d < SODA_HIP_MAX_DIMS; ++d) dims_piece[d] = local_dims[d];
    dims_piece[last] = r1 - r0;
    void* sp = (char*)from + r0 * row_bytes;
    void* dp = (char*)to + r0 * row_bytes;
    plan->out_final_only = final_only ? true : was_final_only;
    const int e = soda_hip_sweep(plan, &sp, &dp, dims_piece, step, lo, hi, stream);
    plan->out_final_only = was_final_only;
    return e;
  };
  while (done < iterate && !rc) {
    if (!pending) {
      rc = exchange((char*)src, recut ? recut_messages((size_t)k) : static_messages());
      count += slab->world > 1;
    }
    if (!rc) rc = ghosts_have_landed();
    if (rc) break;
    pending = false;
    const int step = std::min(slab->exchange, iterate - done);
    // valid region of the slab's input: sides cut inside valid rows are fully valid, the
    // global sides of a static slab carry the margin of the iterations done so far (a
    // re-cut rank's sub-array starts and ends at rows that are valid: every side is cut)
    int32_t lo[SODA_HIP_MAX_DIMS], hi[SODA_HIP_MAX_DIMS];
    output_margins(plan, done, lo, hi);
    if (recut || g.has_lo) lo[last] = 0;
    if (recut || g.has_hi) hi[last] = 0;
    void* dst = cycle[k % 2];
    if (slab->rank == fail_rank && k == fail_at) {
      rc = fail(SODA_HIP_ERR_DEVICE_RUN, "injected failure of rank %d at super-step %d",
                fail_rank, fail_at);
      break;
    }
    const bool more = done + step < iterate;
    if (recut) {
      const RecutStep& st = table.steps[(size_t)k];
      const int64_t reach_lo = (int64_t)step * slab->reach_lo,
                    reach_hi = (int64_t)step * slab->reach_hi;
      auto piece = [&](const Rows& out, bool final_only) -> int {
        return sweep_rows(src, dst, out.lo - reach_lo - table.base, out.hi + reach_hi - table.base,
                          step, lo, hi, final_only);
      };
      std::vector<Rows> bands;
      Rows interior;
      const Rows out{st.cuts[slab->rank], st.cuts[slab->rank + 1]};
      if (overlapped && recut_pieces(table, slab, (size_t)k, &bands, &interior)) {
        for (const Rows& band : bands)
          if (!rc) rc = piece(band, true);
        if (!rc) {
          rc = exchange((char*)dst, recut_messages((size_t)k + 1));   // beside the interior
          count += 1;
          pending = true;
        }
        if (!rc) rc = piece(interior, true);
      } else if (!out.empty()) {
        rc = piece(out, false);
      }
    } else if (overlapped && more && !(g.own < 2 * (send_down + send_up) + 1)) {
      const int64_t first_own = g.ghost_lo, last_own = g.ghost_lo + g.own;
      const int64_t reach_lo = (int64_t)step * slab->reach_lo,
                    reach_hi = (int64_t)step * slab->reach_hi;
      auto piece = [&](int64_t r0, int64_t r1, bool cut_lo, bool cut_hi) -> int {
        int32_t plo[SODA_HIP_MAX_DIMS], phi[SODA_HIP_MAX_DIMS];
        for (int d = 0; d < SODA_HIP_MAX_DIMS; ++d) { plo[d] = lo[d]; phi[d] = hi[d]; }
        if (cut_lo) plo[last] = 0;
        if (cut_hi) phi[last] = 0;
        return sweep_rows(src, dst, r0, r1, step, plo, phi, true);
      };
      int64_t lo_edge = first_own, hi_edge = last_own;
      if (g.has_lo) {    // the lower neighbour's ghost rows: our first send_down rows
        rc = piece(first_own - reach_lo, first_own + send_down + reach_hi, true, true);
        lo_edge = first_own + send_down;
      }
      if (!rc && g.has_hi) {
        rc = piece(last_own - send_up - reach_lo, last_own + reach_hi, true, true);
        hi_edge = last_own - send_up;
      }
      if (!rc) {
        rc = exchange((char*)dst, static_messages());        // beside the interior sweep
        count += 1;
        pending = true;
      }
      if (!rc)
        rc = piece(g.has_lo ? lo_edge - reach_lo : 0,
                   g.has_hi ? hi_edge + reach_hi : g.extent, g.has_lo, g.has_hi);
    } else {
      rc = soda_hip_sweep(plan, &src, &dst, local_dims, step, lo, hi, stream);
    }
    src = dst;
    done += step;
    ++k;
  }
  if (!rc) rc = ghosts_have_landed();
  if (rc) return give_up(rc);
  *result = src;
  if (exchanges) *exchanges = count;
  return 0;
}

// how many (kernel, box) pairs of the plan run a MEASURED (chunk, workgroups per CU)
// instead of the kernel's calibrated one (soda_hip_plan_tune's streaming step)
int soda_hip_plan_tuned_streams(const soda_hip_plan* plan, int* n) {
  if (!plan || !n) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  *n = (int)plan->tuned_stream.size();
  return 0;
}

// ------------------------------------------------- host-buffer entry point
int soda_hip_run_buffers(soda_hip_plan* plan, soda_hip_buffer_t* const* inputs,
                         soda_hip_buffer_t* const* outputs, int iterate,
                         soda_hip_timing* timing) {
  if (!plan || !inputs || !outputs) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "NULL argument");
  const soda_hip_program& p = plan->prog;
  for (int j = 0; j < p.n_inputs; ++j)
    if (!inputs[j]) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "input buffer %d is NULL", j);
  for (int j = 0; j < p.n_outputs; ++j)
    if (!outputs[j]) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "output buffer %d is NULL", j);
  int32_t mlo[SODA_HIP_MAX_DIMS], mhi[SODA_HIP_MAX_DIMS];
  if (iterate < 1) return fail(SODA_HIP_ERR_CONSTRAINT, "iterate must be >= 1");
  output_margins(plan, iterate, mlo, mhi);

  // Bounds-query mode (host.py:204-252): a buffer with neither host nor device
  // memory only gets the shape it must have written into it, and nothing runs.
  // As the generated reference code (its halide_rewrite_buffer, host.py:100-113,
  // sets min / extent / stride of all four dimensions and leaves elem_size
  // alone): a null OUTPUT keeps its min and extents and gets dense strides; a
  // null INPUT gets the first output's min and that output's extents plus the
  // stencil window minus one, the window being the one between the FIRST input and
  // the first output (core.get_stencil_dim(get_overall_stencil_window(input 0,
  // output 0)), host.py:226-233) - for every null input, as in the reference.
  bool query = false;
  auto is_null = [](const soda_hip_buffer_t* b) { return b->host == nullptr && b->dev == 0; };
  for (int j = 0; j < p.n_outputs; ++j) query |= is_null(outputs[j]);
  for (int j = 0; j < p.n_inputs; ++j) query |= is_null(inputs[j]);
  if (query) {
    const soda_hip_buffer_t* o0 = outputs[0];
    // As the reference (host.py:226-233): the window between the FIRST input and the
    // first output, whichever input is asked about - for denoise2d / denoise3d, whose
    // first input `f` is read at the cell itself only, that is too small for `u`; it
    // is the reference's answer all the same.  The composed window of `iterate`
    // iterations with only input 0 as origin:
    Box window{};
    {
      const int nt = n_tensors(p);
      std::vector<Box> feed(p.n_inputs, Box{});
      feed[0].set = true;
      std::vector<Box> cur;
      for (int it = 0; it < iterate; ++it) {
        cur.assign(nt, Box{});
        for (int i = 0; i < p.n_inputs; ++i) cur[i] = feed[i];
        for (int st = 0; st < p.n_stages; ++st) {
          const int t = p.n_inputs + st;
          Box acc{};
          for (int w = 0; w < p.n_windows; ++w) {
            const soda_hip_window& win = p.window[w];
            if (win.stage != t || !cur[win.parent].set) continue;
            const Box& par = cur[win.parent];
            for (int d = 0; d < p.dim; ++d) {
              const int32_t lo = par.lo[d] + win.lo[d], hi = par.hi[d] + win.hi[d];
              acc.lo[d] = acc.set ? std::min(acc.lo[d], lo) : lo;
              acc.hi[d] = acc.set ? std::max(acc.hi[d], hi) : hi;
            }
            acc.set = true;
          }
          for (int d = 0; d < p.dim && acc.set; ++d) {   // boxes contain the cell itself
            acc.lo[d] = std::min<int32_t>(acc.lo[d], 0);
            acc.hi[d] = std::max<int32_t>(acc.hi[d], 0);
          }
          cur[t] = acc;
        }
        if (p.n_inputs == p.n_outputs)
          for (int j = 0; j < p.n_inputs; ++j) feed[j] = cur[p.output_tensor[j]];
      }
      window = cur[p.output_tensor[0]];
      if (!window.set)      // the first output does not depend on the first input
        window = plan->boxes[iterate - 1][p.output_tensor[0]];
    }
    for (int j = 0; j < p.n_outputs; ++j) {
      soda_hip_buffer_t* b = outputs[j];
      if (!is_null(b)) continue;
      int32_t stride = 1;
      for (int d = 0; d < 4; ++d) {
        if (d < p.dim) { b->stride[d] = stride; stride *= b->extent[d]; }
        else { b->min[d] = b->extent[d] = b->stride[d] = 0; }
      }
    }
    for (int j = 0; j < p.n_inputs; ++j) {
      soda_hip_buffer_t* b = inputs[j];
      if (!is_null(b)) continue;
      int32_t stride = 1;
      for (int d = 0; d < 4; ++d) {
        if (d < p.dim) {
          b->min[d] = o0->min[d];
          b->extent[d] = o0->extent[d] + window.hi[d] - window.lo[d];
          b->stride[d] = stride;
          stride *= b->extent[d];
        } else {
          b->min[d] = b->extent[d] = b->stride[d] = 0;
        }
      }
    }
    return 0;
  }

  // element-size checks (host.py:254-255, :969-982)
  for (int j = 0; j < p.n_outputs; ++j)
    if (outputs[j]->elem_size != p.elem_size[p.output_tensor[j]]) {
      fprintf(stderr, "Buffer output %d has elem_size %d instead of %d\n", j,
              outputs[j]->elem_size, p.elem_size[p.output_tensor[j]]);
      return fail(SODA_HIP_ERR_BAD_ELEM_SIZE, "output %d: elem_size %d, expected %d", j,
                  outputs[j]->elem_size, p.elem_size[p.output_tensor[j]]);
    }
  for (int j = 0; j < p.n_inputs; ++j)
    if (inputs[j]->elem_size != p.elem_size[j]) {
      fprintf(stderr, "Buffer input %d has elem_size %d instead of %d\n", j,
              inputs[j]->elem_size, p.elem_size[j]);
      return fail(SODA_HIP_ERR_BAD_ELEM_SIZE, "input %d: elem_size %d, expected %d", j,
                  inputs[j]->elem_size, p.elem_size[j]);
    }
  int64_t dims[SODA_HIP_MAX_DIMS] = {1, 1, 1, 1};
  size_t cells = 1;
  for (int d = 0; d < p.dim; ++d) {
    dims[d] = inputs[0]->extent[d];
    if (dims[d] <= 0) return fail(SODA_HIP_ERR_CONSTRAINT, "extent[%d] = %lld", d,
                                  (long long)dims[d]);
    cells *= (size_t)dims[d];
  }
  auto dense = [&](const soda_hip_buffer_t* b, const char* what, int j) -> int {
    int64_t stride = 1;
    for (int d = 0; d < p.dim; ++d) {
      if (b->extent[d] != dims[d])
        return fail(SODA_HIP_ERR_CONSTRAINT, "%s %d: extent[%d] = %d, expected %lld", what,
                    j, d, b->extent[d], (long long)dims[d]);
      if (b->stride[d] != stride)
        return fail(SODA_HIP_ERR_CONSTRAINT, "%s %d: stride[%d] = %d, expected %lld "
                    "(dense row-major arrays only)", what, j, d, b->stride[d],
                    (long long)stride);
      stride *= dims[d];
    }
    if (!b->host) return fail(SODA_HIP_ERR_NULL_ARGUMENT, "%s %d has no host memory", what, j);
    return 0;
  };
  for (int j = 0; j < p.n_inputs; ++j) { int rc = dense(inputs[j], "input", j); if (rc) return rc; }
  for (int j = 0; j < p.n_outputs; ++j) { int rc = dense(outputs[j], "output", j); if (rc) return rc; }

  std::vector<void*> din(p.n_inputs, nullptr), dout(p.n_outputs, nullptr);
  int rc = 0;
  auto cleanup = [&]() {
    for (void* q : din) if (q) (void)hipFree(q);
    for (void* q : dout) if (q) (void)hipFree(q);
  };
  for (int j = 0; j < p.n_inputs && !rc; ++j) {
    const size_t bytes = cells * p.elem_size[j];
    if (hipMalloc(&din[j], bytes) != hipSuccess)
      rc = fail(SODA_HIP_ERR_DEVICE_MALLOC, "hipMalloc(%zu) failed", bytes);
    else if (hipMemcpy(din[j], inputs[j]->host, bytes, hipMemcpyHostToDevice) != hipSuccess)
      rc = fail(SODA_HIP_ERR_COPY_TO_DEVICE, "H2D copy of input %d failed", j);
  }
  for (int j = 0; j < p.n_outputs && !rc; ++j) {
    const size_t bytes = cells * p.elem_size[p.output_tensor[j]];
    if (hipMalloc(&dout[j], bytes) != hipSuccess)
      rc = fail(SODA_HIP_ERR_DEVICE_MALLOC, "hipMalloc(%zu) failed", bytes);
    else if (hipMemset(dout[j], 0, bytes) != hipSuccess)
      rc = fail(SODA_HIP_ERR_DEVICE_RUN, "hipMemset failed");
  }
  soda_hip_timing local;
  if (!rc) rc = soda_hip_sweep_timed(plan, din.data(), dout.data(), dims, iterate, 1, 1,
                                     nullptr, &local);
  if (!rc) {
    // host.py:796-800: pixels = product of input extents, not multiplied by iterate
    printf("Kernel execution time: %lf us\n", local.kernel_us);
    printf("Kernel throughput: %lf pixel/ns\n", (double)cells / local.kernel_us / 1e3);
    fflush(stdout);
    if (timing) *timing = local;
  }
  // only the valid interior goes back to the caller (host.py:838-899)
  for (int j = 0; j < p.n_outputs && !rc; ++j) {
    const int es = p.elem_size[p.output_tensor[j]];
    int64_t lo[4] = {0, 0, 0, 0}, hi[4] = {1, 1, 1, 1}, ext[4] = {1, 1, 1, 1};
    bool empty = false;
    // each output has its own composed window (host.py:1082-1091)
    const Box& ob = plan->boxes[iterate - 1][p.output_tensor[j]];
    for (int d = 0; d < p.dim; ++d) {
      lo[d] = -ob.lo[d]; hi[d] = dims[d] - ob.hi[d]; ext[d] = dims[d];
      if (hi[d] <= lo[d]) empty = true;
    }
    if (empty) continue;
    // one 3-D copy per index of the fourth dimension
    for (int64_t w = lo[3]; w < hi[3] && !rc; ++w) {
      const size_t skip = (size_t)w * ext[0] * ext[1] * ext[2] * es;
      hipMemcpy3DParms parms;
      memset(&parms, 0, sizeof parms);
      parms.srcPtr = make_hipPitchedPtr((char*)dout[j] + skip, ext[0] * es, ext[0] * es,
                                        ext[1]);
      parms.dstPtr = make_hipPitchedPtr((char*)outputs[j]->host + skip, ext[0] * es,
                                        ext[0] * es, ext[1]);
      parms.srcPos = make_hipPos(lo[0] * es, lo[1], lo[2]);
      parms.dstPos = make_hipPos(lo[0] * es, lo[1], lo[2]);
      parms.extent = make_hipExtent((hi[0] - lo[0]) * es, hi[1] - lo[1], hi[2] - lo[2]);
      parms.kind = hipMemcpyDeviceToHost;
      if (hipMemcpy3D(&parms) != hipSuccess)
        rc = fail(SODA_HIP_ERR_COPY_TO_HOST, "D2H copy of output %d failed: %s", j,
                  hipGetErrorString(hipGetLastError()));
    }
  }
  cleanup();
  return rc;
}

}  // extern "C"
