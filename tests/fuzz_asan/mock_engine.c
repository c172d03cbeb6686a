/* A device stand-in for sanitizer runs of the host side's THREADED paths (api.c: parser pool, batch calls, two-step pulls): h264bsdInit()
 * binds to it like to the HIP engine, frame jobs are swallowed, and a "picture" is eight bytes — the running picture number of the
 * job that was decoded into the frame buffer — so that a harness can tell which picture a pull handed out.  No pixels, no GPU.
 * TEST INFRASTRUCTURE (tests/fuzz_asan/batch_tsan.c); the product links engine.hip instead. */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "engine.h"
#include "framejob.h"

typedef struct Mock { uint64_t *mirror[FJ_MAX_SLOTS + 1]; uint32_t n_slots; int out_slot; long submitted; } Mock;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;     /* stands for the engine's lock */
static long g_jobs;

static int m_configure(void *u, uint32_t wmb, uint32_t hmb, uint32_t n_slots)
{
    Mock *m = (Mock *)u; (void)wmb; (void)hmb;
    pthread_mutex_lock(&g_mu);
    for (uint32_t i = 0; i <= FJ_MAX_SLOTS; i++) { free(m->mirror[i]); m->mirror[i] = NULL; }
    m->n_slots = n_slots;
    for (uint32_t i = 0; i < n_slots && i <= FJ_MAX_SLOTS; i++) m->mirror[i] = (uint64_t *)calloc(1, sizeof(uint64_t));
    pthread_mutex_unlock(&g_mu);
    return 0;
}
static int m_submit(void *u, const uint8_t *blob, uint32_t bytes)
{
    Mock *m = (Mock *)u;
    const FjHeader *h = (const FjHeader *)blob;
    if (bytes < sizeof(FjHeader) || h->cur_slot >= m->n_slots || !m->mirror[h->cur_slot]) return -1;
    if (!h->dbk_only) *m->mirror[h->cur_slot] = h->pic_seq;
    m->submitted++;
    pthread_mutex_lock(&g_mu); g_jobs++; pthread_mutex_unlock(&g_mu);
    return 0;
}
static int m_fetch_begin(void *u, uint32_t slot)
{
    Mock *m = (Mock *)u;
    m->out_slot = -1;
    if (slot >= m->n_slots || !m->mirror[slot]) return -1;
    pthread_mutex_lock(&g_mu); pthread_mutex_unlock(&g_mu);
    m->out_slot = (int)slot;
    return 0;
}
static uint8_t *m_fetch_end(void *u) { Mock *m = (Mock *)u; return m->out_slot < 0 ? NULL : (uint8_t *)m->mirror[m->out_slot]; }
static uint8_t *m_fetch(void *u, uint32_t slot) { return m_fetch_begin(u, slot) ? NULL : m_fetch_end(u); }
static void m_close(void *u)
{
    Mock *m = (Mock *)u;
    for (uint32_t i = 0; i <= FJ_MAX_SLOTS; i++) free(m->mirror[i]);
    free(m);
}
static uint32_t m_errors(void *u) { (void)u; return 0; }

int eng_attach(JobSink *s)
{
    Mock *m = (Mock *)calloc(1, sizeof(Mock));
    if (!m) return -1;
    m->out_slot = -1;
    s->user = m; s->configure = m_configure; s->submit = m_submit; s->fetch = m_fetch; s->fetch_begin = m_fetch_begin; s->fetch_end = m_fetch_end;
    s->close = m_close; s->errors = m_errors;
    return 0;
}
void eng_convert_host(int f, uint32_t w, uint32_t h, const uint8_t *d, uint32_t *o) { (void)f; (void)w; (void)h; (void)d; (void)o; }
int eng_sink_device(const JobSink *s) { (void)s; return 0; }
int eng_device_cpus(int d, int *c, int m) { (void)d; (void)c; (void)m; return 0; }
long mock_jobs(void) { pthread_mutex_lock(&g_mu); const long n = g_jobs; pthread_mutex_unlock(&g_mu); return n; }
