/* Sampling profiler of the integration library's threads (measurement aid, off unless started):
   a process CPU-time timer (ITIMER_PROF) interrupts whichever thread is burning CPU and the
   handler records the interrupted program counter.  tools/host_profile.py starts it around a
   step of the live encoder on the GPU box and maps the samples to symbols with nm - the host
   profile of the real seam with the real device feed, which gprof of a static host-only build
   cannot give (no perf on the boxes). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <signal.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>

static uintptr_t *g_pc;
static long g_cap;
static volatile long g_n;

static void prof_tick(int sig, siginfo_t *si, void *uc) {
  long i;
  (void)sig;
  (void)si;
  i = __sync_fetch_and_add(&g_n, 1);
  if (i < g_cap) g_pc[i] = (uintptr_t)((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP];
}

/* Starts sampling every `usec` microseconds of process CPU time into a buffer of `cap` samples. */
int od_hipenc_prof_start(long cap, int usec) {
  struct sigaction sa;
  struct itimerval it;
  if (cap < 1 || usec < 50) return -1;
  free(g_pc);
  g_pc = (uintptr_t *)calloc(cap, sizeof(*g_pc));
  if (g_pc == NULL) return -1;
  g_cap = cap;
  g_n = 0;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = prof_tick;
  sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigemptyset(&sa.sa_mask);
  if (sigaction(SIGPROF, &sa, NULL) != 0) return -1;
  it.it_interval.tv_sec = 0;
  it.it_interval.tv_usec = usec;
  it.it_value = it.it_interval;
  return setitimer(ITIMER_PROF, &it, NULL);
}

/* Stops sampling; copies up to `cap` program counters to `out`, returns how many ticks fired;
   *base receives the load address of this library (for nm offsets). */
long od_hipenc_prof_stop(uintptr_t *out, long cap, uintptr_t *base) {
  struct itimerval it;
  Dl_info di;
  long n;
  memset(&it, 0, sizeof(it));
  setitimer(ITIMER_PROF, &it, NULL);
  signal(SIGPROF, SIG_IGN);
  n = g_n < g_cap ? g_n : g_cap;
  if (out != NULL) memcpy(out, g_pc, sizeof(*out)*(size_t)(n < cap ? n : cap));
  if (base != NULL) *base = dladdr((void *)od_hipenc_prof_stop, &di) ? (uintptr_t)di.dli_fbase : 0;
  return g_n;
}
