"""C++ host printer: emits the translation unit that DEFINES the generated
entry points for C/C++ callers, on top of libsoda_hip.so.

Counterpart of the reference's host printer as a whole (reference
src/soda/codegen/xilinx/host.py:1169-1206): the emitted file gives

  extern "C" int <app>(buffer_t* in..., buffer_t* out..., const char* blob);
  extern "C" int <app>_test(const char* blob, const int dims[4]);

with the reference's protocol -- `<app>` checks, runs and writes back the valid
interior (host.py:186-945, done by soda_hip_run_buffers), `<app>_test` fills the
reference's input pattern, calls `<app>`, recomputes on the CPU with the plain
loop nest and compares (host.py:984-1167).  What the reference bakes in as
OpenCL boiler-plate is here a handful of descriptor literals for the C ABI.
"""
from . import spec as specmod
from .kernel_common import program_hash, tensor_index

_COORD = 'pqrs'


def _array(values):
  return '{%s}' % ', '.join(str(int(v)) for v in values)


HEADERS = ('cassert cfloat cmath cstdbool cstddef cstdint cstdio cstdlib cstring '
           'algorithm array string unordered_map').split()


def print_prologue(out):
  """Same system headers as the reference's host translation unit
  (host.py:12-36): they decide which overload an unqualified `sqrt(float)` is
  (the C `double sqrt(double)`)."""
  for h in HEADERS:
    out.write('#include <%s>\n' % h)
  out.write('static FILE* const* error_report = &stderr;\n')


def print_selfcheck(spec, out):
  """`<app>_selfcheck`: the CPU half of `<app>_test` (reference
  host.py:1073-1146): recompute `iterate` iterations with the plain loop nest,
  compare the device outputs on the box the reference compares (exact for
  integers, squared relative error for floats), report up to `max_report`
  mismatches, return their count.  Reads the device result, never produces
  one."""
  app = spec['app_name']
  dim = spec['dim']
  types = specmod.tensor_c_types(spec)
  index = tensor_index(spec)
  ins = [t['name'] for t in spec['inputs']]
  outs = list(spec['outputs'])
  stages = spec['stages']
  windows = []
  for stage, wins in specmod.stage_windows(spec).items():
    for parent, (lo, hi) in wins.items():
      windows.append((index[stage], index[parent], lo, hi))
  w = out.write
  w('extern "C" long long %s_selfcheck(const void* const* inputs, '
    'const void* const* device_outputs, const int dims[4], int iterate, '
    'double threshold, int max_report) {\n' % app)
  w('  long long error_count = 0;\n')
  w('  (void)threshold; (void)max_report;\n')
  w('  size_t cells = 1;\n  for (int d = 0; d < %d; ++d) cells *= (size_t)dims[d];\n'
    % dim)
  w('  const int64_t stride0 = 1; (void)stride0;\n')
  for d in range(1, dim):
    w('  const int64_t stride%d = %s;\n' % (d, ' * '.join(
        '(int64_t)dims[%d]' % x for x in range(d))))
  w('  static const int32_t win[][%d] = {%s};\n' % (2 + 2 * dim, ', '.join(
      _array([st, pa] + list(lo) + list(hi)) for st, pa, lo, hi in windows)))
  w('  int32_t box_lo[%d][%d], box_hi[%d][%d];  // per tensor, composed window\n'
    % (len(index), dim, len(index), dim))
  w('  memset(box_lo, 0, sizeof box_lo); memset(box_hi, 0, sizeof box_hi);\n')
  for s in stages:
    n = s['name']
    for k in range(2 if n in outs else 1):
      w('  %s* buf_%s_%d = new %s[cells]();\n' % (types[n], n, k, types[n]))
  w('  for (int it = 0; it < iterate; ++it) {\n')
  w('    const bool last = it == iterate - 1;\n')
  w('    for (int s = %d; s < %d; ++s) {\n' % (len(ins), len(index)))
  w('      bool first = true;\n')
  w('      for (size_t k = 0; k < sizeof win / sizeof win[0]; ++k) {\n')
  w('        if (win[k][0] != s) continue;\n')
  w('        for (int d = 0; d < %d; ++d) {\n' % dim)
  w('          const int32_t lo = box_lo[win[k][1]][d] + win[k][2 + d];\n')
  w('          const int32_t hi = box_hi[win[k][1]][d] + win[k][%d + d];\n' % (2 + dim))
  w('          box_lo[s][d] = first ? lo : std::min(box_lo[s][d], lo);\n')
  w('          box_hi[s][d] = first ? hi : std::max(box_hi[s][d], hi);\n')
  w('        }\n        first = false;\n      }\n')
  w('      for (int d = 0; d < %d; ++d) {  // the cell itself is inside the array\n'
    % dim)
  w('        box_lo[s][d] = std::min(box_lo[s][d], 0);\n')
  w('        box_hi[s][d] = std::max(box_hi[s][d], 0);\n')
  w('      }\n    }\n')
  for j, n in enumerate(ins):
    if len(ins) == len(outs):
      # output j of the previous iteration feeds input j (core.py:342-360)
      w('    const %s* %s_img = it == 0 ? (const %s*)inputs[%d] : (((it - 1) & 1) ? '
        'buf_%s_1 : buf_%s_0);\n' % (types[n], n, types[n], j, outs[j], outs[j]))
    else:
      w('    const %s* %s_img = (const %s*)inputs[%d];\n' % (types[n], n, types[n], j))
  for s in stages:
    n = s['name']
    if n in outs:
      w('    %s* %s_img = (it & 1) ? buf_%s_1 : buf_%s_0;\n' % (types[n], n, n, n))
    else:
      w('    %s* %s_img = buf_%s_0;\n' % (types[n], n, n))

  def load(name, rel):
    return '%s_img[%s]' % (name, ' + '.join(
        '(%c%+d)*stride%d' % (_COORD[d], rel[d], d) for d in range(dim)))

  for s in stages:
    n = s['name']
    t = index[n]
    w('    // produce %s\n' % n)
    if n in outs:
      w('    const %s* fpga_%s = (const %s*)device_outputs[%d];\n'
        % (types[n], n, types[n], outs.index(n)))
      w('#pragma omp parallel for reduction(+:error_count)\n')
    else:
      w('#pragma omp parallel for\n')
    for d in reversed(range(dim)):
      w('    for (int32_t {v} = -box_lo[{t}][{d}]; {v} < dims[{d}] - box_hi[{t}][{d}]; '
        '++{v})\n'.format(v=_COORD[d], t=t, d=d))
    w('    {\n')
    for let in s['lets']:
      w('      const %s %s = %s;\n' % (let['c_type'], let['name'],
                                      specmod.substitute_loads(let['expr'], load)))
    cell = ' + '.join('%c*stride%d' % (_COORD[d], d) for d in range(dim))
    w('      const %s result = %s;\n' % (types[n], specmod.substitute_loads(
        s['expr'], load)))
    w('      %s_img[%s] = result;\n' % (n, cell))
    if n in outs:
      fmt = ', '.join(['%d'] * dim)
      coords = ', '.join(_COORD[:dim])
      w('      if (last) {\n')
      w('        const %s val_fpga = fpga_%s[%s];\n' % (types[n], n, cell))
      w('        const %s val_cpu = result;\n' % types[n])
      if specmod.is_float_type(s['haoda_type']):
        w('        if (double(val_fpga-val_cpu)*double(val_fpga-val_cpu)/(double('
          'val_cpu)*double(val_cpu)) > threshold * threshold) {\n')
        w('          if (error_count < max_report) fprintf(*error_report, '
          '"%%lf != %%lf @(%s)\\n", double(val_fpga), double(val_cpu), %s);\n'
          % (fmt, coords))
      else:
        w('        if (val_fpga != val_cpu) {\n')
        w('          if (error_count < max_report) fprintf(*error_report, '
          '"%%ld != %%ld @(%s)\\n", (long)val_fpga, (long)val_cpu, %s);\n'
          % (fmt, coords))
      w('          ++error_count;\n        }\n      }\n')
    w('    }\n')
  if len(ins) == len(outs):
    for j, n in enumerate(ins):
      w('    for (int d = 0; d < %d; ++d) { box_lo[%d][d] = box_lo[%d][d]; '
        'box_hi[%d][d] = box_hi[%d][d]; }\n' % (dim, j, index[outs[j]], j,
                                                index[outs[j]]))
  w('  }\n')
  for s in stages:
    n = s['name']
    for k in range(2 if n in outs else 1):
      w('  delete[] buf_%s_%d;\n' % (n, k))
  w('  return error_count;\n}\n\n')


def print_multi_gpu(spec, w, app, name_in, name_out, dim, deepest):
  """`<app>_multi_gpu`: the grid cut into slabs along the outermost dimension,
  all GPUs of the node driven from ONE process, one thread per GPU
  (ncclCommInitAll + soda_hip_run_slab).  Compiled in with -DSODA_HIP_MULTI_GPU
  (needs -lrccl -lpthread).  Same protocol as `<app>`: caller owns the host
  arrays, only the valid interior of the output is written."""
  w('#ifdef SODA_HIP_MULTI_GPU\n#include <rccl/rccl.h>\n#include <thread>\n#include <mutex>\n#include <condition_variable>\n'
    '#include <vector>\n')
  w('extern "C" int %s_multi_gpu(buffer_t* var_%s_buffer, buffer_t* var_%s_buffer, '
    'const char* blob, int iterate, int ngpu) {\n' % (app, name_in, name_out))
  w('  const int dim = %d, last = dim - 1;\n' % dim)
  w('  int count = 0;\n  soda_hip_device_count(&count);\n')
  # rehearsal: every rank on the devices that exist (rank % count).  RCCL refuses two
  # ranks on one GPU, so this only works over a stand-in for it (tests/rccl_standin);
  # it lets the slab logic of this function run on a one-GPU box.
  w('  const bool rehearse = getenv("SODA_HIP_REHEARSE_RANKS_ON_ONE_GPU") != nullptr;\n')
  w('  if (count < 1) return SODA_HIP_ERR_NO_DEVICE;\n')
  w('  if (ngpu > count && !rehearse) ngpu = count;\n')
  w('  if (ngpu < 1) return SODA_HIP_ERR_NO_DEVICE;\n')
  w('  buffer_t* in = var_%s_buffer; buffer_t* out = var_%s_buffer;\n'
    % (name_in, name_out))
  w('  const int64_t rows = in->extent[last];\n')
  w('  if (rows < ngpu) ngpu = (int)rows;\n')
  w('  std::vector<ncclComm_t> comms(ngpu, nullptr);\n')
  w('  if (ngpu > 1) {\n    std::vector<int> devs(ngpu);\n'
    '    for (int i = 0; i < ngpu; ++i) devs[i] = i % count;\n'
    '    if (ncclCommInitAll(comms.data(), ngpu, devs.data()) != ncclSuccess) {\n'
    '      fprintf(*error_report, "ERROR: ncclCommInitAll failed\\n");\n'
    '      return SODA_HIP_ERR_NO_DEVICE;\n    }\n  }\n')
  w('  std::vector<int> status(ngpu, 0);\n')
  # Every rank derives the SAME exchange period and the same cuts from global figures
  # before anything is sent; a rank that fails during set-up is seen by all at a
  # rendezvous in front of the first message, and the thread of one that fails later
  # aborts EVERY rank's communicator, once (ncclCommAbort is local to a rank: aborting
  # the peers' own communicators is what takes them out of ncclRecv; the library is told
  # to leave the communicator alone, soda_hip_slab.abort_on_error = 0 - one owner).
  w('  std::mutex gate; std::condition_variable gate_cv; int arrived = 0; '
    'bool setup_failed = false, aborted = false;\n')
  w('  auto rendezvous = [&](bool ok) {\n'
    '    std::unique_lock<std::mutex> lock(gate);\n'
    '    if (!ok) setup_failed = true;\n'
    '    if (++arrived == ngpu) gate_cv.notify_all();\n'
    '    else gate_cv.wait(lock, [&] { return arrived == ngpu; });\n'
    '    return !setup_failed;\n  };\n')
  w('  auto abort_all = [&]() {\n'
    '    std::lock_guard<std::mutex> lock(gate);\n'
    '    if (aborted) return;\n    aborted = true;\n'
    '    for (ncclComm_t& comm : comms) if (comm) { ncclCommAbort(comm); comm = nullptr; }\n'
    '  };\n')
  w('  auto worker = [&](int rank) {\n')
  w('    int& rc = status[rank];\n')
  w('    soda_hip_module* module = nullptr;\n    soda_hip_plan* plan = nullptr;\n')
  w('    void *a = nullptr, *b = nullptr, *c = nullptr, *result = nullptr;\n')
  w('    soda_hip_slab slab;\n    memset(&slab, 0, sizeof slab);\n')
  w('    int64_t local[4] = {1, 1, 1, 1}, in_at = 0, res_first = 0, res_last = 0, '
    'res_at = 0, own = 0;\n')
  w('    size_t row_bytes = 0;\n')
  w('    rc = soda_hip_set_device(rank % count);\n')
  w('    if (!rc) rc = soda_hip_module_load_file(blob, &module);\n')
  w('    soda_hip_program program;\n    fill_program(&program);\n')
  w('    soda_hip_kernel kernels[32];\n    const int n_kernels = fill_kernels(kernels);\n')
  w('    if (!rc) rc = soda_hip_plan_create(module, &program, kernels, n_kernels, &plan);\n')
  w('    if (rc == 0) {\n')
  w('      int32_t lo[4], hi[4];\n      soda_hip_plan_margins(plan, 1, lo, hi);\n')
  w('      slab.rank = rank; slab.world = ngpu;\n')
  w('      slab.reach_lo = lo[last]; slab.reach_hi = hi[last];\n')
  w('      for (int d = 0; d < dim; ++d) slab.dims[d] = in->extent[d];\n')
  w('      const int64_t base = rows / ngpu, extra = rows % ngpu;\n')
  w('      slab.own_first = rank * base + (rank < extra ? rank : extra);\n')
  w('      slab.own_last = slab.own_first + base + (rank < extra ? 1 : 0);\n')
  # exchange period: the rule of runtime/dist.py: auto_exchange
  w('      const int reach = lo[last] > hi[last] ? lo[last] : hi[last];\n')
  w('      int exchange = %d * 8;   // eight launches of the deepest fused kernel\n'
    % deepest)
  w('      while (exchange > %d && (int64_t)exchange * (reach > 0 ? reach : 1) * 2 * 100 > '
    'base * 15) exchange -= %d;\n' % (deepest, deepest))
  w('      if (exchange > iterate) exchange = iterate;\n')
  w('      if (exchange < 1) exchange = 1;\n')
  # slabs re-cut to the shrinking valid box every super-step (include/soda_hip.h:
  # SODA_HIP_SLAB_CUT_RECUT) unless the environment asks for the static cut
  w('      slab.cut = getenv("SODA_HIP_SLAB_STATIC_CUT") ? SODA_HIP_SLAB_CUT_STATIC : '
    'SODA_HIP_SLAB_CUT_RECUT;\n')
  w('      slab.abort_on_error = 0;   // abort_all below owns every communicator\n')
  w('      // static cut: never deeper than the smallest slab (all ranks compute the same value)\n')
  w('      if (slab.cut == SODA_HIP_SLAB_CUT_STATIC)\n')
  w('        rc = soda_hip_slab_exchange(rows, ngpu, lo[last], hi[last], exchange, &exchange);\n')
  w('      slab.exchange = ngpu > 1 ? exchange : iterate;\n')
  # the order of exchange and sweeps: serial unless the environment asks for the
  # bands-first order (include/soda_hip.h: SODA_HIP_SLAB_BANDS_FIRST)
  w('      slab.order = getenv("SODA_HIP_SLAB_BANDS_FIRST") ? SODA_HIP_SLAB_BANDS_FIRST : '
    'SODA_HIP_SLAB_SERIAL;\n')
  w('      if (!rc) rc = soda_hip_slab_layout(plan, &slab, iterate, local, &in_at, &res_first, '
    '&res_last, &res_at);\n')
  w('      row_bytes = (size_t)in->elem_size;\n')
  w('      for (int d = 0; d < last; ++d) row_bytes *= (size_t)in->extent[d];\n')
  w('      const size_t bytes = row_bytes * (size_t)local[last];\n')
  w('      own = slab.own_last - slab.own_first;\n')
  w('      if (!rc) rc = soda_hip_malloc(&a, bytes);\n')
  w('      if (!rc) rc = soda_hip_malloc(&b, bytes);\n')
  w('      if (!rc) rc = soda_hip_malloc(&c, bytes);\n')
  w('      if (!rc) rc = soda_hip_memset(a, 0, bytes, nullptr);\n')
  w('      if (!rc) rc = soda_hip_memset(b, 0, bytes, nullptr);\n')
  w('      if (!rc) rc = soda_hip_memset(c, 0, bytes, nullptr);\n')
  w('      if (!rc) rc = soda_hip_memcpy_h2d((char*)a + in_at * row_bytes, '
    'in->host + slab.own_first * row_bytes, own * row_bytes, nullptr);\n')
  w('      if (!rc) rc = soda_hip_stream_synchronize(nullptr);\n')
  w('    }\n')
  w('    if (rc) fprintf(*error_report, "ERROR: GPU %d: %s: %s\\n", rank, '
    'soda_hip_error_name(rc), soda_hip_last_error());\n')
  w('    // nobody sends before everybody is ready to receive\n')
  w('    const bool go = rendezvous(rc == 0);\n')
  w('    if (!go && !rc) rc = SODA_HIP_ERR_GENERIC;   // a peer failed during set-up\n')
  w('    if (go) {\n')
  w('      rc = soda_hip_run_slab(plan, &slab, comms[rank], a, b, c, iterate, '
    'nullptr, &result, nullptr);\n')
  w('      if (!rc) rc = soda_hip_stream_synchronize(nullptr);\n')
  w('      if (rc) {\n')
  w('        fprintf(*error_report, "ERROR: GPU %d: %s: %s\\n", rank, '
    'soda_hip_error_name(rc), soda_hip_last_error());\n')
  w('        if (ngpu > 1) abort_all();   // peers blocked in ncclRecv return with an error\n')
  w('      }\n')
  w('      if (!rc) {\n')
  w('        // only the valid interior goes back to the caller (host.py:838-899)\n')
  w('        // the rows of the result this rank holds (a re-cut run: its share of the rows '
    'still valid)\n')
  w('        const int64_t held = res_last - res_first;\n')
  w('        std::vector<uint8_t> stage((size_t)(held > 0 ? held : 0) * row_bytes + 1);\n')
  w('        if (held > 0) rc = soda_hip_memcpy_d2h(stage.data(), (char*)result + res_at * '
    'row_bytes, held * row_bytes, nullptr);\n')
  w('        if (!rc) rc = soda_hip_stream_synchronize(nullptr);\n')
  w('        int32_t mlo[4], mhi[4];\n')
  w('        soda_hip_plan_margins(plan, iterate, mlo, mhi);\n')
  w('        const int64_t es = in->elem_size;\n')
  w('        const int64_t x0 = mlo[0], x1 = in->extent[0] - mhi[0];\n')
  w('        const int64_t inner_rows = row_bytes / (in->extent[0] * es);\n')
  w('        for (int64_t y = res_first; !rc && x1 > x0 && y < res_last; ++y) {\n')
  w('          if (y < mlo[last] || y >= rows - mhi[last]) continue;\n')
  w('          for (int64_t q = 0; q < inner_rows; ++q) {\n')
  if dim == 3:
    w('            if (q < mlo[1] || q >= in->extent[1] - mhi[1]) continue;\n')
  if dim == 4:
    w('            const int64_t q1 = q % in->extent[1], q2 = q / in->extent[1];\n')
    w('            if (q1 < mlo[1] || q1 >= in->extent[1] - mhi[1] || q2 < mlo[2] || '
      'q2 >= in->extent[2] - mhi[2]) continue;\n')
  w('            const size_t off = (size_t)(q * in->extent[0] + x0) * es;\n')
  w('            memcpy(out->host + y * row_bytes + off, stage.data() + '
    '(y - res_first) * row_bytes + off, (size_t)(x1 - x0) * es);\n')
  w('          }\n        }\n      }\n    }\n')
  w('    soda_hip_free(a); soda_hip_free(b); soda_hip_free(c);\n')
  w('    soda_hip_plan_destroy(plan);\n    soda_hip_module_unload(module);\n')
  w('  };\n')
  w('  std::vector<std::thread> threads;\n')
  w('  for (int r = 0; r < ngpu; ++r) threads.emplace_back(worker, r);\n')
  w('  for (auto& t : threads) t.join();\n')
  w('  for (ncclComm_t comm : comms) if (comm) ncclCommDestroy(comm);\n')
  w('  for (int rc : status) if (rc) return rc;\n')
  w('  return 0;\n}\n#endif  // SODA_HIP_MULTI_GPU\n\n')


def print_code(spec, kernels, out, lowered=None):
  """`spec`: the source program (golden loops, entry points); `lowered`: the
  program the kernels were generated from (descriptor literals)."""
  lowered = spec if lowered is None else lowered
  app = spec['app_name']
  dim = spec['dim']
  types = specmod.tensor_c_types(spec)
  index = tensor_index(spec)
  ins = [t['name'] for t in spec['inputs']]
  outs = list(spec['outputs'])
  stages = spec['stages']
  windows = []
  for stage, wins in specmod.stage_windows(spec).items():
    for parent, (lo, hi) in wins.items():
      windows.append((index[stage], index[parent], lo, hi))
  w = out.write
  w('// Host program for SODA kernel `%s`, generated by sodac (HIP back end).\n' % app)
  w('// Program hash %s.  Build: g++ -std=c++11 -fopenmp -ffp-contract=off -Iinclude\n'
    % program_hash(spec))
  w('//   %s_host.cpp -Lsoda-compiler_amd/csrc -lsoda_hip\n' % app)
  print_prologue(out)
  w('#include "soda_hip.h"\n\n')
  w('typedef soda_hip_buffer_t buffer_t;   // layout of reference header.py:36-48\n')
  w('static const char kProgramHash[] = "%s";\n' % program_hash(spec))
  w('static const int kIterate = %d;\n\n' % spec['iterate'])

  # ---- descriptors (of the LOWERED program the kernels implement) ------------
  low_index = tensor_index(lowered)
  low_types = specmod.tensor_c_types(lowered)
  low_windows = []
  for stage, wins in specmod.stage_windows(lowered).items():
    for parent, (lo, hi) in wins.items():
      low_windows.append((low_index[stage], low_index[parent], lo, hi))
  w('static void fill_program(soda_hip_program* p) {\n')
  w('  memset(p, 0, sizeof *p);\n')
  w('  p->dim = %d; p->n_inputs = %d; p->n_stages = %d; p->n_outputs = %d;\n'
    % (dim, len(ins), len(lowered['stages']), len(outs)))
  sizes = [0] * len(low_index)
  for n, i in low_index.items():
    sizes[i] = specmod.ELEM_SIZE[low_types[n]]
  w('  static const int32_t elem[] = %s;\n' % _array(sizes))
  w('  for (int i = 0; i < %d; ++i) p->elem_size[i] = elem[i];\n' % len(sizes))
  for j, o in enumerate(outs):
    w('  p->output_tensor[%d] = %d;\n' % (j, low_index[o]))
  w('  p->n_windows = %d;\n' % len(low_windows))
  for k, (st, pa, lo, hi) in enumerate(low_windows):
    w('  { soda_hip_window* v = &p->window[%d]; v->stage = %d; v->parent = %d;' % (
        k, st, pa))
    for d in range(dim):
      w(' v->lo[%d] = %d; v->hi[%d] = %d;' % (d, lo[d], d, hi[d]))
    w(' }\n')
  w('}\n\n')
  w('static int fill_kernels(soda_hip_kernel* k) {\n')
  w('  memset(k, 0, sizeof(soda_hip_kernel) * %d);\n' % len(kernels))
  for i, kd in enumerate(kernels):
    w('  snprintf(k[%d].name, sizeof k[%d].name, "%s"); k[%d].kind = %s; '
      'k[%d].depth = %d; k[%d].stage = %d; k[%d].fill_rows = %d; '
      'k[%d].origin_align = %d; k[%d].min_extent[0] = %d; k[%d].min_extent[1] = %d; '
      'k[%d].step_valu = %d; k[%d].step_bytes = %d; k[%d].step_ns_full = %d; '
      'k[%d].step_ns_one = %d; k[%d].stream_gbps = %d; k[%d].xcd_tiles = %d; '
      'k[%d].stream_wgs_per_cu = %d; k[%d].fade_lo_mib = %d; k[%d].fade_hi_mib = %d; '
      'k[%d].stream_chunk = %d; k[%d].edge_slack = %d;\n' % (
          i, i, kd['name'], i,
          'SODA_HIP_KERNEL_FUSED' if kd['kind'] == 'fused' else 'SODA_HIP_KERNEL_STAGE',
          i, kd['depth'], i, kd['stage'], i, kd.get('fill_rows', 0),
          i, kd.get('origin_align', 0), i, kd.get('min_extent', [0, 0])[0],
          i, kd.get('min_extent', [0, 0])[1], i, kd.get('step_valu', 0),
          i, kd.get('step_bytes', 0), i, kd.get('step_ns_full', 0),
          i, kd.get('step_ns_one', 0), i, kd.get('stream_gbps', 0),
          i, kd.get('xcd_tiles', 0), i, kd.get('stream_wgs_per_cu', 0),
          i, kd.get('fade_lo_mib', 0), i, kd.get('fade_hi_mib', 0),
          i, kd.get('stream_chunk', 0), i, kd.get('edge_slack', 0)))
    w('  { static const int32_t b[] = %s, t[] = %s; for (int d = 0; d < 3; ++d) '
      'k[%d].block[d] = b[d]; for (int d = 0; d < 4; ++d) k[%d].tile[d] = t[d]; }\n'
      % (_array(kd['block']), _array(kd['tile']), i, i))
  w('  return %d;\n}\n\n' % len(kernels))

  # ---- <app> -------------------------------------------------------------------
  params = ''.join('buffer_t* var_%s_buffer, ' % n for n in ins + outs)
  w('static int %s_iterate(%sconst char* blob, int iterate) {\n' % (app, params))
  w('  soda_hip_module* module = nullptr;\n  soda_hip_plan* plan = nullptr;\n')
  w('  int rc = soda_hip_module_load_file(blob, &module);\n')
  w('  if (rc != 0) { fprintf(*error_report, "ERROR: %s\\n", soda_hip_last_error());'
    ' return rc; }\n')
  w('  static char meta[1 << 20];\n  size_t meta_len = 0;\n')
  w('  soda_hip_module_meta(module, meta, sizeof meta, &meta_len);\n')
  w('  if (strstr(meta, kProgramHash) == nullptr) {\n')
  w('    fprintf(*error_report, "ERROR: %s was not generated for this program\\n", '
    'blob);\n    soda_hip_module_unload(module);\n    return SODA_HIP_ERR_MISMATCH;\n  }\n')
  w('  soda_hip_program program;\n  fill_program(&program);\n')
  w('  soda_hip_kernel kernels[%d];\n  const int n_kernels = fill_kernels(kernels);\n'
    % max(1, len(kernels)))
  w('  rc = soda_hip_plan_create(module, &program, kernels, n_kernels, &plan);\n')
  w('  if (rc == 0) {\n')
  w('    buffer_t* inputs[] = {%s};\n' % ', '.join('var_%s_buffer' % n for n in ins))
  w('    buffer_t* outputs[] = {%s};\n' % ', '.join('var_%s_buffer' % n for n in outs))
  w('    rc = soda_hip_run_buffers(plan, inputs, outputs, iterate, nullptr);\n')
  w('  }\n')
  w('  if (rc != 0) fprintf(*error_report, "ERROR: %s: %s\\n", '
    'soda_hip_error_name(rc), soda_hip_last_error());\n')
  w('  soda_hip_plan_destroy(plan);\n  soda_hip_module_unload(module);\n')
  w('  return rc;\n}\n\n')
  w('extern "C" int %s(%sconst char* blob) {\n' % (app, params))
  w('  return %s_iterate(%sblob, kIterate);\n}\n\n' % (
      app, ''.join('var_%s_buffer, ' % n for n in ins + outs)))

  # ---- <app>_multi_gpu ---------------------------------------------------------
  if len(ins) == 1 and len(outs) == 1:
    print_multi_gpu(spec, w, app, ins[0], outs[0], dim, max(
        [k['depth'] for k in kernels if k['kind'] == 'fused'] or [1]))

  # ---- <app>_test --------------------------------------------------------------
  print_selfcheck(spec, out)
  w('extern "C" int %s_test(const char* blob, const int dims[4]) {\n' % app)
  w('  const int iterate = getenv("SODA_ITERATE") ? atoi(getenv("SODA_ITERATE")) : '
    'kIterate;\n')
  w('  size_t cells = 1;\n  for (int d = 0; d < %d; ++d) cells *= (size_t)dims[d];\n'
    % dim)
  w('  const int64_t stride0 = 1; (void)stride0;\n')
  for d in range(1, dim):
    w('  const int64_t stride%d = %s;\n' % (d, ' * '.join(
        '(int64_t)dims[%d]' % x for x in range(d))))
  for n in ins + outs:
    w('  buffer_t %s; memset(&%s, 0, sizeof(buffer_t));\n' % (n, n))
    w('  %s* %s_host = new %s[cells]();\n' % (types[n], n, types[n]))
    for d in range(dim):
      w('  %s.extent[%d] = dims[%d]; %s.stride[%d] = (int32_t)stride%d;\n'
        % (n, d, d, n, d, d))
    w('  %s.elem_size = sizeof(%s); %s.host = (uint8_t*)%s_host;\n'
      % (n, types[n], n, n))
  # reference init pattern (host.py:1033-1051)
  floaty = specmod.is_float_type(spec['inputs'][0]['haoda_type'])
  for n in ins:
    w('#pragma omp parallel for\n')
    for d in reversed(range(dim)):
      w('  for (int32_t {v} = 0; {v} < dims[{d}]; ++{v})\n'.format(v=_COORD[d], d=d))
    summed = '+'.join(_COORD[:dim])
    value = summed
    if floaty:
      value = '%s(%s)/%s(%s)' % (types[n], summed, types[n],
                                 '+'.join('dims[%d]' % d for d in range(dim)))
    w('    %s_host[%s] = %s;\n' % (n, ' + '.join(
        '%c*stride%d' % (_COORD[d], d) for d in range(dim)), value))
  if len(ins) == 1 and len(outs) == 1:
    w('#ifdef SODA_HIP_MULTI_GPU\n')
    w('  const int run_rc = getenv("SODA_GPUS") ? %s_multi_gpu(&%s, &%s, blob, iterate, '
      'atoi(getenv("SODA_GPUS"))) : %s_iterate(&%s, &%s, blob, iterate);\n'
      % (app, ins[0], outs[0], app, ins[0], outs[0]))
    w('#else\n')
  w('  const int run_rc = %s_iterate(%sblob, iterate);\n' % (
      app, ''.join('&%s, ' % n for n in ins + outs)))
  if len(ins) == 1 and len(outs) == 1:
    w('#endif\n')
  w('  if (run_rc != 0) return run_rc < 0 ? -run_rc : run_rc;\n')
  w('  double threshold = 0.00001;\n')
  w('  if (nullptr != getenv("THRESHOLD")) threshold = atof(getenv("THRESHOLD"));\n')
  w('  const void* in_ptrs[] = {%s};\n' % ', '.join('%s_host' % n for n in ins))
  w('  const void* out_ptrs[] = {%s};\n' % ', '.join('%s_host' % n for n in outs))
  w('  const int error_count = (int)%s_selfcheck(in_ptrs, out_ptrs, dims, iterate, '
    'threshold, 32);\n' % app)
  w('  fprintf(*error_report, error_count == 0 ? "INFO: PASS!\\n" : '
    '"INFO: FAIL!\\n");\n')
  for n in ins + outs:
    w('  delete[] %s_host;\n' % n)
  w('  return error_count;\n}\n\n')

  # ---- the README test-bench (README.md:76-91) ---------------------------------
  w('#ifdef SODA_HIP_MAIN\n')
  w('int main(int argc, char** argv) {\n')
  w('  if (argc != %d) {\n' % (2 + dim))
  w('    fprintf(stderr, "Usage: \\n    %%s <blob> %s\\n", argv[0]);\n' % ' '.join(
      '<extent %d>' % d for d in range(dim)))
  w('    return 1;\n  }\n')
  w('  int dims[4] = {0, 0, 0, 0};\n')
  w('  for (int d = 0; d < %d; ++d) dims[d] = atoi(argv[2 + d]);\n' % dim)
  w('  return %s_test(argv[1], dims) == 0 ? 0 : 1;\n}\n' % app)
  w('#endif  // SODA_HIP_MAIN\n')
