/* CPU oracle, timing helper.  TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg).
 *
 * Pins the threads of the process's OpenMP team, one per CPU of a list, so that the
 * baseline's `#pragma omp parallel for` loops (the reference's host.py:1081) run on
 * the same cores from sample to sample: under a cgroup CPU quota on a large host the
 * threads otherwise wander over every socket between parallel regions, and the pages
 * the parallel first touch placed stay behind.  libgomp keeps its team between
 * parallel regions of the same size, so the affinity set here holds for the
 * oracle's loops that follow.  cpus[t] < 0 leaves thread t alone. */
#define _GNU_SOURCE
#include <omp.h>
#include <sched.h>

int soda_oracle_pin(const int* cpus, int n) {
  int failed = 0;
#pragma omp parallel num_threads(n) reduction(+ : failed)
  {
    const int t = omp_get_thread_num();
    if (t < n && cpus[t] >= 0) {
      cpu_set_t set;
      CPU_ZERO(&set);
      CPU_SET(cpus[t], &set);
      if (sched_setaffinity(0, sizeof set, &set) != 0) failed += 1;
    }
  }
  return failed;
}

/* back to a whole mask (the caller's own thread when the sample is over) */
int soda_oracle_unpin(const int* cpus, int n_cpus, int threads) {
  int failed = 0;
#pragma omp parallel num_threads(threads) reduction(+ : failed)
  {
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int i = 0; i < n_cpus; ++i) CPU_SET(cpus[i], &set);
    if (sched_setaffinity(0, sizeof set, &set) != 0) failed += 1;
  }
  return failed;
}
