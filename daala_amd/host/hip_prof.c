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

#if defined(__x86_64__)
static uintptr_t *g_pc;
static long g_cap;
static volatile long g_n;
static volatile int g_on;            /* the timer is armed */
static struct sigaction g_prev;      /* the embedding application's SIGPROF disposition */

static void prof_tick(int sig, siginfo_t *si, void *uc) {
  long i;
  (void)sig;
  (void)si;
  if (!g_on) return;
  i = __sync_fetch_and_add(&g_n, 1);
  if (i < g_cap) g_pc[i] = (uintptr_t)((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP];
}

static void prof_disarm(void) {
  struct itimerval it;
  memset(&it, 0, sizeof(it));
  setitimer(ITIMER_PROF, &it, NULL);
  g_on = 0;
  __sync_synchronize();
}

/* Starts sampling every `usec` microseconds of process CPU time into a buffer of `cap` samples.
   A second start without a stop re-arms with a fresh buffer (the timer is stopped first: no
   tick can see a buffer being replaced). */
int od_hipenc_prof_start(long cap, int usec) {
  struct sigaction sa;
  struct itimerval it;
  uintptr_t *buf;
  int was_on;
  if (cap < 1 || usec < 50) return -1;
  was_on = g_on;
  prof_disarm();
  buf = (uintptr_t *)calloc(cap, sizeof(*buf));
  if (buf == NULL) return -1;
  free(g_pc);
  g_pc = buf;
  g_cap = cap;
  g_n = 0;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = prof_tick;
  sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigemptyset(&sa.sa_mask);
  if (sigaction(SIGPROF, &sa, was_on ? NULL : &g_prev) != 0) return -1;
  g_on = 1;
  it.it_interval.tv_sec = 0;
  it.it_interval.tv_usec = usec;
  it.it_value = it.it_interval;
  return setitimer(ITIMER_PROF, &it, NULL);
}

/* Stops sampling and puts the previous SIGPROF disposition back; copies up to `cap` program
   counters to `out`, returns how many ticks fired; *base receives the load address of this
   library (for nm offsets). */
long od_hipenc_prof_stop(uintptr_t *out, long cap, uintptr_t *base) {
  Dl_info di;
  long n;
  int was_on;
  was_on = g_on;
  prof_disarm();
  if (was_on) sigaction(SIGPROF, &g_prev, NULL);
  n = g_n < g_cap ? g_n : g_cap;
  if (out != NULL && g_pc != NULL) memcpy(out, g_pc, sizeof(*out)*(size_t)(n < cap ? n : cap));
  if (base != NULL) *base = dladdr((void *)od_hipenc_prof_stop, &di) ? (uintptr_t)di.dli_fbase : 0;
  return g_n;
}
#else
/* the interrupted program counter is read from the x86-64 signal context: no profiler elsewhere */
int od_hipenc_prof_start(long cap, int usec) {
  (void)cap;
  (void)usec;
  return -1;
}
long od_hipenc_prof_stop(uintptr_t *out, long cap, uintptr_t *base) {
  (void)out;
  (void)cap;
  (void)base;
  return -1;
}
#endif
