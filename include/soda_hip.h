/* soda_hip.h -- C ABI of libsoda_hip.so, the MI355X (gfx950) run-time of the
 * SODA HIP back end.
 *
 * The reference compiler (UCLA-VAST/soda-compiler) has no run-time library: its
 * host printer emits a whole OpenCL host program per stencil
 * (src/soda/codegen/xilinx/host.py).  The pieces of that generated program that
 * sit on the stencil hot path are what this library replaces; each entry point
 * cites the generated function (by the printer lines that emit it) it stands in
 * for.  `sodac --hip-host` emits a thin shim that only fills in the descriptors
 * below and calls these functions (INTEGRATION.md shows the shim).
 *
 * Conventions
 *   - plain C, no HIP/torch types: device pointers are `void*`, a stream is a
 *     `void*` holding a hipStream_t (NULL = the null stream);
 *   - every function returns 0 on success or a negative code; nothing calls
 *     exit() (the generated reference host does, e.g. host.py:399-404).  Codes
 *     reuse the reference's Halide numbering (host.py:118-133) where one applies;
 *   - soda_hip_last_error() gives the detail text of the calling thread's most
 *     recent failure;
 *   - dimension 0 is the fastest-varying one (stride 1), the LAST dimension is
 *     the streamed / outermost one (host.py:1014-1017).
 */
#ifndef SODA_HIP_H_
#define SODA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SODA_HIP_ABI_VERSION 8
#define SODA_HIP_MAX_DIMS 4
#define SODA_HIP_MAX_TENSORS 16 /* inputs + stages of one program */
#define SODA_HIP_MAX_IO 8
#define SODA_HIP_MAX_WINDOWS 64
#define SODA_HIP_MAX_KERNELS 32

/* ---- return codes (reference host.py:118-133 numbering) ------------------ */
enum {
  SODA_HIP_OK = 0,
  SODA_HIP_ERR_GENERIC = -1,
  SODA_HIP_ERR_BAD_ELEM_SIZE = -3,       /* halide_error_code_bad_elem_size */
  SODA_HIP_ERR_OUT_OF_BOUNDS = -4,       /* ..._access_out_of_bounds */
  SODA_HIP_ERR_EXTENTS_TOO_LARGE = -6,   /* ..._buffer_extents_too_large */
  SODA_HIP_ERR_CONSTRAINT = -8,          /* ..._constraint_violated */
  SODA_HIP_ERR_OUT_OF_MEMORY = -11,      /* ..._out_of_memory */
  SODA_HIP_ERR_NULL_ARGUMENT = -12,      /* ..._buffer_argument_is_null */
  SODA_HIP_ERR_COPY_TO_HOST = -14,       /* ..._copy_to_host_failed */
  SODA_HIP_ERR_COPY_TO_DEVICE = -15,     /* ..._copy_to_device_failed */
  SODA_HIP_ERR_DEVICE_MALLOC = -16,      /* ..._device_malloc_failed */
  SODA_HIP_ERR_DEVICE_SYNC = -17,        /* ..._device_sync_failed */
  SODA_HIP_ERR_DEVICE_FREE = -18,        /* ..._device_free_failed */
  SODA_HIP_ERR_NO_DEVICE = -19,          /* ..._no_device_interface */
  SODA_HIP_ERR_INTERNAL = -22,           /* ..._internal_error */
  SODA_HIP_ERR_DEVICE_RUN = -23,         /* ..._device_run_failed */
  /* no Halide counterpart */
  SODA_HIP_ERR_COMPILE = -100,           /* hiprtc rejected the kernel text */
  SODA_HIP_ERR_MODULE = -101,            /* code object unreadable / wrong arch */
  SODA_HIP_ERR_NO_KERNEL = -102,         /* kernel name not in the module */
  SODA_HIP_ERR_MISMATCH = -103           /* blob built for another program */
};

const char* soda_hip_error_name(int code);
const char* soda_hip_last_error(void);
int soda_hip_abi_version(void);

/* ---- device ---------------------------------------------------------------
 * Replaces the platform/device/context discovery of the generated host
 * (host.py:412-560). */
int soda_hip_device_count(int* count);
int soda_hip_set_device(int ordinal);
/* name: gcnArchName such as "gfx950:sramecc+:xnack-"; cu: compute units */
int soda_hip_device_info(int ordinal, char* name, size_t name_cap, int* cu,
                         uint64_t* total_mem_bytes);
int soda_hip_malloc(void** dev, size_t bytes);           /* host.py:574-594 */
int soda_hip_free(void* dev);                            /* host.py:913-924 */
int soda_hip_memset(void* dev, int value, size_t bytes, void* stream);
int soda_hip_memcpy_h2d(void* dev, const void* host, size_t bytes, void* stream);
int soda_hip_memcpy_d2h(void* host, const void* dev, size_t bytes, void* stream);
int soda_hip_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
int soda_hip_stream_synchronize(void* stream);           /* host.py:781-790 */

/* ---- kernel blob ------------------------------------------------------------
 * The `xclbin` argument of the generated entry points (host.py:931, :992)
 * becomes a "blob": either a gfx950 code object built ahead of time from the
 * text `sodac --hip-kernel` prints, or that text itself, compiled here with
 * hiprtc (replaces load_xclbin2_to_memory + clCreateProgramWithBinary,
 * host.py:49-97, :520-560). */
typedef struct soda_hip_module soda_hip_module;

int soda_hip_module_load_file(const char* code_object_path,
                              soda_hip_module** module);
int soda_hip_module_load_data(const void* image, size_t bytes,
                              soda_hip_module** module);
/* arch NULL = the current device's; options are extra hiprtc flags */
int soda_hip_module_compile(const char* source, const char* arch,
                            const char* const* options, int n_options,
                            soda_hip_module** module);
/* the compiled code object, e.g. to cache it on disk */
int soda_hip_module_image(const soda_hip_module* module, const void** image,
                          size_t* bytes);
/* JSON text the kernel printer stored in the blob (`soda_hip_meta` symbol):
 * program hash, spec and the kernel table.  Copies up to cap-1 bytes. */
int soda_hip_module_meta(const soda_hip_module* module, char* buf, size_t cap,
                         size_t* length);
int soda_hip_module_unload(soda_hip_module* module);

/* ---- program + kernel descriptors ----------------------------------------
 * What the reference bakes into the generated host as literals: tensor types
 * (host.py:254-255), the stencil window (STENCIL_DIM_n, host.py:1183-1197).
 * Tensors are numbered inputs first, then stages in execution order. */
typedef struct soda_hip_window {
  int32_t stage;   /* tensor index of the reading stage */
  int32_t parent;  /* tensor index of the tensor read */
  int32_t lo[SODA_HIP_MAX_DIMS];  /* hull of (load index - store index) */
  int32_t hi[SODA_HIP_MAX_DIMS];
} soda_hip_window;

typedef struct soda_hip_program {
  int32_t dim;
  int32_t n_inputs;
  int32_t n_stages;
  int32_t n_outputs;
  int32_t elem_size[SODA_HIP_MAX_TENSORS];
  int32_t output_tensor[SODA_HIP_MAX_IO]; /* tensor index of output j; it feeds
                                             input j when iterating */
  int32_t n_windows;
  soda_hip_window window[SODA_HIP_MAX_WINDOWS];
} soda_hip_program;

enum {
  SODA_HIP_KERNEL_STAGE = 0, /* one stage of one iteration, result to HBM */
  SODA_HIP_KERNEL_FUSED = 1  /* `depth` whole iterations, all stages fused */
};

typedef struct soda_hip_kernel {
  char name[96];
  int32_t kind;
  int32_t depth;    /* FUSED: iterations advanced per launch */
  int32_t stage;    /* STAGE: tensor index produced */
  int32_t block[3]; /* workgroup shape */
  int32_t tile[SODA_HIP_MAX_DIMS]; /* output cells one workgroup produces; for a
                                      streaming kernel the outer-dimension entry
                                      is only the default chunk length */
  int32_t fill_rows; /* streaming kernels: extra outer-dimension rows a workgroup
                        walks through before its first output row (pipeline
                        fill + halo); 0 = not a streaming kernel */
  int32_t origin_align; /* > 1: the kernel starts its dimension-0 tiles at box_lo[0]
                           rounded down to a multiple of this (cache-line aligned
                           strips) and tile[0] is exact; 0/1: tiles start at
                           box_lo[0] (tile[0] already allows for the kernel's own
                           rounding) */
  int32_t min_extent[2]; /* > 0: the kernel handles only arrays with dims[0] and
                            dims[1] at least this large (its tiles are moved
                            inside the array rather than guarded) and fewer than
                            2^30 cells per plane (32-bit in-plane offsets); the
                            scheduler skips it otherwise */
  /* Cost figures of a streaming kernel, from the kernel printer (0 = none): the
   * scheduler prices every fused depth with them and splits `iterate` into the
   * cheapest sequence of launches (soda_hip.cpp: step_seconds). */
  int32_t step_valu;  /* VALU issue cycles ONE workgroup (all its wavefronts
                         together) spends per streamed row / plane */
  int32_t step_bytes; /* HBM bytes one workgroup loads + stores per step */
  /* MEASURED figures of this kernel (tools/calibrate.py on an MI355X, kept in
   * soda_hip/codegen/calibration.json and carried in the blob's metadata; all 0 =
   * not calibrated: the scheduler prices the kernel with step_valu / step_bytes): */
  int32_t step_ns_full; /* nanoseconds per streamed step with the chip full of this
                           kernel's workgroups on arrays that fit the Infinity Cache
                           (what the kernel's own pipeline costs) */
  int32_t step_ns_one;  /* the same with at most one workgroup per CU */
  int32_t stream_gbps;  /* GB/s of step_bytes the kernel sustains on arrays far larger
                           than the caches.  A PRICE constant, not a physical bandwidth:
                           the price charges (chunk + fill rows) steps of step_bytes each,
                           while a short chunk skips the loads past its last row and never
                           stores during fill, so with a measured stream_chunk the figure
                           can exceed the 8 TB/s of HBM and is valid at THAT chunk length
                           only (the launcher uses it there and nowhere else) */
  int32_t xcd_tiles;  /* N > 0: the kernel takes a 1-D grid and places its tiles
                         itself, XCD by XCD (N = most tiles per super-tile; 1 =
                         the plain round-robin deal): the launcher cuts the plane of tiles
                         (dimensions 0 and 1) into super-tiles of SX x SY tiles,
                         passes param[1] = SX | SY << 16 and param[2] = (super-
                         tiles along x) | (along y) << 16, and launches
                         ceil(super-tiles x chunks / 8) x 8 x SX x SY workgroups.
                         N < 0: a 1-D grid of 8 P workgroups, P = ceil(tiles / 8) in
                         param[3]; workgroup L works on tile (L mod 8) P + L / 8
                         of the x-fastest tile order (XCD L mod 8 takes a RUN of
                         consecutive tiles); param[1] = 1 | 1 << 16, param[2] =
                         (tiles along x) | (along y) << 16 */
  /* (ABI 5) */
  int32_t stream_wgs_per_cu; /* N > 0: a launch whose box (inputs + outputs) does not
                         fit the Infinity Cache keeps at most N workgroups of this kernel
                         on a CU (the launcher asks for dynamic LDS the kernel never
                         touches) and walks correspondingly longer chunks: memory-bound
                         kernels move more bytes per second when the chip streams a few
                         dozen row ranges than when it streams hundreds (tools/copyceil.hip;
                         with the seam-free depth-1 strips: jacobi2d 16384^2 -4 %, sobel2d
                         -6 % under the bench protocol).  0 = no cap */
  int32_t fade_lo_mib;  /* footprints (MiB of the launch's box, inputs + outputs) between */
  int32_t fade_hi_mib;  /* which the HBM term of the price (stream_gbps) fades in: below
                         fade_lo the arrays live in the Infinity Cache and a step costs
                         step_ns_*, above fade_hi the kernel streams.  Part of the
                         kernel's calibration record (tools/calibrate.py); 0, 0 = the
                         defaults 128 and 512 */
  /* (ABI 6) */
  int32_t stream_chunk; /* N > 0: a launch whose box does not fit the Infinity Cache walks
                         chunks of N rows (planes) per workgroup instead of the longest
                         chunk that fills the chip in whole rounds.  Measured per kernel
                         (tools/calibrate.py: the fastest of a few chunk lengths and caps on
                         a streaming array; stream_wgs_per_cu and stream_gbps then belong
                         to THIS chunk): the memory-bound kernels move up to 15 % more
                         bytes per second when many short-lived workgroups, dispatched in
                         address order, keep the rows the chip works on close together
                         (jacobi2d 16384^2 depth 1: 436 -> 378 us with 16 rows, blur 226
                         -> 200, sobel2d 246 -> 212; profiles/r04_stream_chunk.txt).
                         0 = the launcher's own choice */
  /* (ABI 8) */
  int32_t edge_slack;   /* N > 0 (kernels with origin_align > 1 and xcd_tiles < 0: the 3-D
                         block form): a tile computes N valid columns more than the
                         tile[0] it stores (what rounding its width down to origin_align
                         left over), and the FIRST and LAST tile of a row of tiles store
                         them: the launcher starts the tiles at
                           x0 = (box_lo[0] + N) rounded down to origin_align
                         (up to N columns INSIDE the box: the first tile, its window moved N
                         columns to the left, stores [box_lo[0], x0 + tile[0])), launches
                           nx = max(1, ceil((box_hi[0] - x0 - N) / tile[0]))
                         tiles along x (the last one stores up to box_hi[0] <= its start +
                         tile[0] + N) and passes x0 in the upper 32 bits of param[1].  One
                         tile that would have to stretch both ways (nx == 1 and box_hi[0] >
                         x0 + tile[0]), or a box that ends before x0, starts at box_lo[0]
                         rounded down instead.  A box of
                         456 columns takes 4 tiles of 112 instead of 5 (jacobi3d 512^3 x200:
                         boxes 464, 456, 336, 328, 240, 232 and 112 lose a tile column).  0 = tiles start
                         at box_lo[0] rounded down and every tile stores tile[0] columns */
} soda_hip_kernel;

/* By-value argument of every generated kernel. */
typedef struct soda_hip_args {
  void* tensor[SODA_HIP_MAX_TENSORS]; /* by tensor index; unused = NULL */
  int64_t dims[SODA_HIP_MAX_DIMS];    /* array extents */
  int64_t box_lo[SODA_HIP_MAX_DIMS];  /* cells to produce: [box_lo, box_hi) */
  int64_t box_hi[SODA_HIP_MAX_DIMS];
  int64_t param[4]; /* param[0]: outer-dimension rows per workgroup, chosen per
                       launch so that the grid fills the chip in whole rounds;
                       param[1], param[2], param[3]: see soda_hip_kernel.xcd_tiles */
} soda_hip_args;

/* ---- plan -------------------------------------------------------------------
 * A program bound to a blob.  Owns its scratch device memory (ping-pong partner
 * of the outputs, stage intermediates), sized on first use, freed on destroy.
 * Threading (reference: one host thread, one in-order queue, host.py:513): every
 * entry point is synchronous on the host; a plan is NOT re-entrant - its scratch
 * arrays and window tables are shared by its sweeps, so use one plan per host
 * thread and per stream (plans are cheap; modules can be shared).  The error text
 * of soda_hip_last_error() is per thread. */
typedef struct soda_hip_plan soda_hip_plan;

int soda_hip_plan_create(soda_hip_module* module, const soda_hip_program* program,
                         const soda_hip_kernel* kernels, int n_kernels,
                         soda_hip_plan** plan);
int soda_hip_plan_destroy(soda_hip_plan* plan);

/* margins after `iterations` iterations: outputs are defined on
 * [lo[d], dims[d] - hi[d]) (reference core.py:794-835, host.py:1082-1091) */
int soda_hip_plan_margins(const soda_hip_plan* plan, int iterations,
                          int32_t lo[SODA_HIP_MAX_DIMS],
                          int32_t hi[SODA_HIP_MAX_DIMS]);

typedef struct soda_hip_timing {
  double kernel_us;     /* device time of the timed sweep loop (hipEvents) */
  int32_t launches;     /* kernel launches in one sweep loop */
  int32_t max_depth;    /* deepest temporal block used */
  double dominant_us;   /* device time of the dominant kernel's launches in ONE sweep
                           loop, every launch at its fastest of the repeats ... */
  int32_t dominant_launches; /* ... over this many launches (of one sweep loop) */
  char dominant_name[96];
  double fastest_us;    /* all launches of one sweep loop, each at its fastest repeat
                           (per-launch events add gaps: compare with the wall time of
                           an un-instrumented loop before quoting a kernel time) */
} soda_hip_timing;

/* The device sweep: `iterate` applications of the program on device arrays.
 *   in[j]   level-0 arrays, never written
 *   out[j]  receive level `iterate`; cells outside the valid box are
 *           unspecified (the reference leaves them unspecified too) and MAY BE
 *           WRITTEN: launches ping-pong through out[j], and a kernel may store the
 *           cells between the box and the next 64-byte boundary of a row along
 *           (whole-line writes); nothing outside the array is ever touched
 *   valid_lo/valid_hi  NULL, or per-dimension margins of the region of `in`
 *           that holds defined data: [valid_lo[d], dims[d] - valid_hi[d]).
 *           NULL means the whole array (a fresh run).  A caller that resumes
 *           after t iterations passes the margins of iteration t; a caller that
 *           owns a slab of a larger grid passes 0 on the sides whose ghost rows
 *           a neighbour just refreshed.
 * The outputs are then defined on the box shrunk by soda_hip_plan_margins(
 * iterate).  Stands in for the clEnqueueTask of the generated host
 * (host.py:775-790) plus the FPGA kernel itself (hls_kernel.py:12-103).
 * Asynchronous on `stream`. */
int soda_hip_sweep(soda_hip_plan* plan, void* const* in, void* const* out,
                   const int64_t dims[SODA_HIP_MAX_DIMS], int iterate,
                   const int32_t* valid_lo, const int32_t* valid_hi,
                   void* stream);

/* Same, bracketed by hipEvents on `stream`: `warmup` untimed runs, then
 * `repeats` timed ones (reference protocol host.py:775-796: one warm-up, one
 * timed run).  Synchronises the stream. */
int soda_hip_sweep_timed(soda_hip_plan* plan, void* const* in, void* const* out,
                         const int64_t dims[SODA_HIP_MAX_DIMS], int iterate,
                         int warmup, int repeats, void* stream,
                         soda_hip_timing* timing);

/* The launch list soda_hip_sweep would issue for these arguments, without
 * running or allocating anything: kernel_index[i] = index into the plan's kernel
 * table of launch i (empty boxes are not launched and not listed), est_us[i] =
 * its modelled duration (0 when the kernel carries no cost figures).  At most
 * `capacity` entries are written; *n_launches is the full count.  For callers
 * that report or plan around the schedule (bench.py, the multi-GPU driver). */
int soda_hip_plan_schedule(soda_hip_plan* plan,
                           const int64_t dims[SODA_HIP_MAX_DIMS], int iterate,
                           const int32_t* valid_lo, const int32_t* valid_hi,
                           int32_t* kernel_index, double* est_us, int capacity,
                           int* n_launches);

/* Restricts fused kernels to depth <= max_depth (0 = no limit); for tests and
 * tuning. */
int soda_hip_plan_set_max_depth(soda_hip_plan* plan, int max_depth);

/* Optional: finds the split of `iterate` into fused depths that runs fastest on THIS
 * device for arrays of these extents.  The scheduler's model ranks depths within a few
 * percent of each other; this takes its split and the ones it gives when each deep
 * kernel in turn is priced 12 % lower or higher, runs every distinct candidate as a
 * whole sweep (in -> out, as soda_hip_sweep; one untimed run, then the faster of two)
 * and remembers the fastest: later sweeps and schedules with the same dims and
 * `iterate` use it.  A split can only change speed, never results.
 * Then the MEMORY-BOUND launches of that schedule (kernels that carry a measured chunk
 * length, soda_hip_kernel.stream_chunk, on boxes beyond the Infinity Cache): the
 * calibrated chunk length x cap on workgroups per CU was measured on one box, and boxes
 * differ by more than the gain - so the calibrated pair is timed against the chunk's two
 * neighbours on the calibration ladder (8, 12, 16, 24, 32, 48, ... rows) and the other
 * cap, whole sweeps again, and a pair that beats it by more than 1 % is kept per (kernel,
 * box extents).  Chunking cannot change results either.  Synchronous; a few dozen
 * sweeps' worth of time (the untimed warm-up run of the reference protocol,
 * host.py:775-790, is the natural place for it). */
int soda_hip_plan_tune(soda_hip_plan* plan, void* const* in, void* const* out,
                       const int64_t dims[SODA_HIP_MAX_DIMS], int iterate,
                       const int32_t* valid_lo, const int32_t* valid_hi, void* stream);

/* Fixes the split of `iterate` into fused depths for arrays of these extents (what
 * soda_hip_plan_tune finds by measurement): depths[0..n) in launch order, each the depth
 * of a fused kernel of the blob, adding up to `iterate`; n = 0 gives the choice back to
 * the scheduler.  Profiling passes use it to repeat the schedule of an earlier run
 * (bench.py --split).  Same-depth kernels are still chosen per launch; a kernel the
 * array is too small for makes the schedule fall back to the scheduler's own. */
int soda_hip_plan_set_split(soda_hip_plan* plan, const int64_t dims[SODA_HIP_MAX_DIMS],
                            int iterate, const int32_t* depths, int n_depths);

/* on != 0: a sweep writes `out` with its LAST launch only; the launches before it
 * alternate between two plan-owned arrays (the second one is allocated when a sweep
 * first needs it).  By default the intermediate launches alternate between one
 * plan-owned array and `out`, i.e. `out` also receives the larger boxes of earlier
 * levels.  A caller that sweeps SUB-ARRAYS of one output array piece by piece (the
 * overlapped multi-GPU schedule: boundary bands first, interior later,
 * soda_hip/runtime/dist.py) needs this: an intermediate box of one piece would
 * reach into rows another piece has already finished. */
int soda_hip_plan_set_out_final_only(soda_hip_plan* plan, int on);

/* *n = how many (kernel, box) pairs of the plan run a (chunk length, workgroups per CU)
 * that soda_hip_plan_tune MEASURED to beat the kernel's calibrated pair on this device by
 * more than 1 % (0: every streaming launch runs its calibration record).  For bench
 * lines: which of the two a timed run used (ABI 8). */
int soda_hip_plan_tuned_streams(const soda_hip_plan* plan, int* n);

/* ---- shader clock under load ------------------------------------------------
 * The deep kernels are bound by VALU issue, and the clock the chip holds under their
 * load (2.0-2.2 of 2.4 GHz) is one of the factors between their rate and the peak's
 * (bench.py: roofline.shader_clock_ghz).  _start launches ONE wavefront of the blob's
 * `soda_hip_clock_probe` kernel on a stream the plan owns: it sleeps for `spins` x ~8 k
 * shader cycles (~4 us each) beside whatever the caller enqueues next, counting shader
 * cycles (s_memtime) and ticks of the constant 100 MHz clock (s_memrealtime).  _finish
 * waits for it and returns cycles / time.  No reference counterpart (the reference reads
 * OpenCL profiling events, host.py:775-800); SODA_HIP_ERR_NO_KERNEL for blobs built
 * before ABI 7. */
int soda_hip_clock_probe_start(soda_hip_plan* plan, int spins);
int soda_hip_clock_probe_finish(soda_hip_plan* plan, double* shader_ghz, double* seconds);

/* ---- multi-GPU: one slab of a larger grid ------------------------------------
 * The reference has nothing distributed (one FPGA); what must be preserved is its
 * semantics: no boundary condition, the valid box shrinks every iteration
 * (core.py:794-835, host.py:1082-1091), hence an OPEN chain of slabs along the
 * outermost dimension.  One process (or thread) per GPU owns rows
 * [own_first, own_last) of that dimension and keeps `exchange * reach` ghost rows
 * on each side that has a neighbour.  Per super-step of `exchange` iterations:
 * neighbours swap ghost rows (ncclSend / ncclRecv inside one group, on `stream`),
 * then the slab advances `exchange` iterations with soda_hip_sweep, the ghost
 * sides declared valid and the global sides carrying the margin of the iterations
 * done so far.  Same logic as soda_hip/runtime/dist.py (the torch.distributed
 * driver, covered by world-size 2 and 3 tests); this entry exists so that a C or
 * C++ caller - the generated `<app>()` - can shard without Python.  librccl.so is
 * loaded on first use.  Tested with world == 1 on the real library and with world
 * 2 to 4 over a test-only stand-in for librccl.so (ranks = host threads sharing one
 * GPU, tests/rccl_standin): every line below the ABI runs; what has NOT run yet is
 * real RCCL with more than one rank (no multi-GPU box was available). */
typedef struct soda_hip_slab {
  int32_t rank, world;
  int32_t reach_lo, reach_hi; /* per-iteration stencil reach along the outermost
                                 dimension towards lower / higher indices
                                 (soda_hip_plan_margins(plan, 1)) */
  int32_t exchange;           /* iterations between ghost exchanges (>= 1) */
  int64_t dims[SODA_HIP_MAX_DIMS]; /* the GLOBAL grid */
  int64_t own_first, own_last;     /* this rank's rows of the outermost dimension: the
                                      rows of the INPUT it is handed (with
                                      SODA_HIP_SLAB_CUT_RECUT: of the even cut,
                                      rank * rows / world rounded as soda_hip_slab_layout
                                      documents) */
  int32_t order;                   /* SODA_HIP_SLAB_SERIAL or SODA_HIP_SLAB_BANDS_FIRST (ABI 7) */
  int32_t cut;                     /* SODA_HIP_SLAB_CUT_STATIC or _RECUT (ABI 8) */
  int32_t abort_on_error;          /* see soda_hip_run_slab, "Failure" (ABI 8) */
  int32_t reserved;                /* 0 */
} soda_hip_slab;

/* Who owns which rows when (ABI 8).
 *   CUT_STATIC  every rank keeps [own_first, own_last) for the whole run and
 *               `exchange * reach` ghost rows beside them.  The valid box shrinks by the
 *               reach every iteration (core.py:794-835), so the first and last ranks run
 *               out of work while the middle ones keep full slabs: a step costs what the
 *               busiest rank costs - jacobi2d 16384^2 x1000 on 8 ranks 87.8 % of an even
 *               share (ghost rows included), jacobi3d 512^3 x200 on 8 ranks 58 %.
 *   CUT_RECUT   every super-step cuts the rows its OUTPUT level defines,
 *               [(done + step) reach_lo, rows - (done + step) reach_hi), evenly again
 *               (93.9 % and 93.7 % for the two runs above); a rank reads its output rows
 *               widened by step x reach, and what it does not hold of those - ghost rows
 *               and rows that changed owner alike - arrives in the super-step's one group
 *               of sends and receives, from whichever ranks hold it (with thin slabs not
 *               only the neighbours).  Same logic as soda_hip/runtime/dist.py: RecutPlan.
 *               The arrays span everything the rank ever holds (soda_hip_slab_layout). */
#define SODA_HIP_SLAB_CUT_STATIC 0
#define SODA_HIP_SLAB_CUT_RECUT 1

/* How soda_hip_run_slab orders a super-step's exchange against its sweeps (the same two
 * orders as soda_hip/runtime/dist.py: SerialSchedule / StreamSchedule):
 *   SERIAL       exchange on `stream`, then one sweep of the whole slab;
 *   BANDS_FIRST  every super-step but the last first sweeps the two bands of rows the
 *                neighbours are waiting for (sub-arrays of 3 E r rows, cut inside valid
 *                data), hands them to the NEXT super-step's exchange on a second stream
 *                the plan owns (events both ways) and sweeps the interior meanwhile; a
 *                piece writes its destination with its last launch only
 *                (soda_hip_plan_set_out_final_only), so intermediate launches never touch
 *                rows that are being sent.  Slabs too thin for bands (own rows <=
 *                2 x the rows sent) sweep whole and then exchange.  Same results. */
#define SODA_HIP_SLAB_SERIAL 0
#define SODA_HIP_SLAB_BANDS_FIRST 1

/* The exchange period every rank of an even cut uses (soda_hip/runtime/dist.py:
 * SlabPlan applies the same rule): a ghost region cannot be deeper than the smallest
 * slab, floor(rows / world) rows, so
 *   *exchange = world > 1 ? max(1, min(wanted, floor(rows / world) / reach)) : wanted,
 * reach = max(reach_lo, reach_hi, 1).  SODA_HIP_ERR_CONSTRAINT when the smallest slab
 * is thinner than the reach of ONE iteration (no exchange period can work).  Every
 * rank computes this from global figures, so all agree before the first message. */
int soda_hip_slab_exchange(int64_t rows, int world, int reach_lo, int reach_hi,
                           int wanted, int* exchange);

/* extents of the rank's local arrays (own rows + ghost rows) and the ghost depths
 * (CUT_STATIC only: a re-cut run's arrays depend on the iteration count,
 * soda_hip_slab_layout) */
int soda_hip_slab_extent(const soda_hip_plan* plan, const soda_hip_slab* slab,
                         int64_t local_dims[SODA_HIP_MAX_DIMS], int64_t* ghost_lo,
                         int64_t* ghost_hi);

/* Where a run of `iterate` iterations keeps its rows, for either cut (ABI 8):
 *   local_dims     extents of the three arrays a, b, c;
 *   *input_offset  row of the arrays at which the rank's rows of the input,
 *                  [own_first, own_last), go (array `a`);
 *   *result_first, *result_last  the GLOBAL rows of the result the rank holds afterwards
 *                  (CUT_STATIC: its own rows; CUT_RECUT: its share of the even cut of the
 *                  rows still valid after `iterate` iterations - possibly none);
 *   *result_offset the row of the result array at which they start.
 * CUT_RECUT requires own_first / own_last to be the even cut of the whole grid:
 * rank r owns rows [r * base + min(r, extra), ... + base + (r < extra)), base = rows /
 * world, extra = rows % world - every rank derives every other rank's rows from it. */
int soda_hip_slab_layout(const soda_hip_plan* plan, const soda_hip_slab* slab, int iterate,
                         int64_t local_dims[SODA_HIP_MAX_DIMS], int64_t* input_offset,
                         int64_t* result_first, int64_t* result_last,
                         int64_t* result_offset);

/* a: level-0 slab (own rows at the layout's input offset, every other row anything),
 * never written except for the rows it receives; b, c: two more arrays of the same
 * size.  comm: an ncclComm_t of `world` ranks (NULL when world == 1).  *result
 * receives b or c, whichever holds the rank's rows of the result after `iterate`
 * iterations (soda_hip_slab_layout says which rows and where).  Asynchronous on `stream`.
 * Failure: the descriptor, `iterate`, the order and the cut are checked before anything
 * is sent; such an error leaves the communicator alone and usable.  Later failures (a
 * launch, an allocation, an RCCL error) happen while peers may be waiting in ncclRecv
 * for rows this rank will not send any more:
 *   abort_on_error == 0  the error is returned and the communicator left to the caller -
 *     for a caller that drives every rank of the node from one process (the generated
 *     `<app>_multi_gpu`: the thread that sees the error aborts EVERY rank's communicator,
 *     once, which is what unblocks the peers);
 *   abort_on_error != 0  after a failure of this rank's own (anything but an error RCCL
 *     itself reports: that may be the caller's watchdog, or a peer's driver, having aborted
 *     this very communicator - it is then left as it is) this rank's communicator is
 *     aborted (ncclCommAbort) before the error is returned, the error text ends in
 *     "(communicator aborted)", and it must not be used, aborted or destroyed again.  Best effort:
 *     ncclCommAbort is LOCAL to the calling rank - it releases this rank's resources and
 *     fails its own pending operations, it does not wake a peer that spins in an enqueued
 *     receive.  A one-process-per-rank caller still has to watch ncclCommGetAsyncError
 *     (or a timeout) on every rank and abort there too. */
int soda_hip_run_slab(soda_hip_plan* plan, const soda_hip_slab* slab, void* comm,
                      void* a, void* b, void* c, int iterate, void* stream,
                      void** result, int* exchanges);

/* ---- host-buffer entry (the generated `<app>`) ------------------------------
 * Legacy Halide buffer_t, bit-compatible with the struct the reference's
 * header printer emits (header.py:36-48). */
typedef struct soda_hip_buffer_t {
  uint64_t dev;
  uint8_t* host;
  int32_t extent[4];
  int32_t stride[4];
  int32_t min[4];
  int32_t elem_size;
  uint8_t host_dirty;
  uint8_t dev_dirty;
  uint8_t _padding[10 - sizeof(void*)];
} soda_hip_buffer_t;

/* `int <app>(buffer_t* in..., buffer_t* out..., const char* xclbin)`
 * (host.py:931-945 -> :186-929): checks element sizes, bounds-query mode when a
 * buffer has host == NULL && dev == 0 (host.py:204-252), device buffers created
 * and freed inside the call, one warm-up then one timed run, prints
 * "Kernel execution time: %lf us" / "Kernel throughput: %lf pixel/ns" to stdout
 * (host.py:796-800), writes only the valid interior of each output back
 * (host.py:838-899).  timing may be NULL. */
int soda_hip_run_buffers(soda_hip_plan* plan, soda_hip_buffer_t* const* inputs,
                         soda_hip_buffer_t* const* outputs, int iterate,
                         soda_hip_timing* timing);

#ifdef __cplusplus
}
#endif
#endif /* SODA_HIP_H_ */
