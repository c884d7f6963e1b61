/* gpu_sampler.c — clock / power / throttle record of ONE GPU while a bench leg runs (measurement tooling, not product).
 *
 * bench.py's roofline legs quote a kernel duration; the chip's DVFS state during that duration is the missing column
 * (MI355X_MICROARCH.md "DVFS give-back").  Round 4 read /sys/class/drm/card<HIP ordinal>/…/power1_average from a Python
 * thread: on a multi-GPU host the HIP ordinal is not the DRM card number (the driver's box read 97 MHz during a saturated
 * scan), hwmon's power1_average is a slow moving average, and the thread competed with the launch loop for the GIL.
 *
 * This file: the device is found by its PCI address (orama_ctx_pci_bus_id → rsmi_dev_pci_id_get), the firmware's metrics
 * table (sysfs gpu_metrics, decoded by librocm_smi64 for whatever table revision the driver exposes) is sampled by a
 * native thread, and the region is summarised from the table's ACCUMULATORS where it has them — energy counter → mean
 * socket power, PPT / thermal residency counters → share of the region spent power- or thermally-throttled — so the figures
 * do not depend on when the samples fell.
 *
 * build: gcc -O2 -shared -fPIC -I/opt/rocm/include gpu_sampler.c -o libgpu_sampler.so -L/opt/rocm/lib -lrocm_smi64 -lpthread
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <rocm_smi/rocm_smi.h>

#define GS_MAX_SAMPLES 65536

typedef struct {
    int available;          /* 0: nothing could be read */
    uint32_t samples;
    double seconds;         /* first to last sample */
    double gfxclk_mhz_median, gfxclk_mhz_min, gfxclk_mhz_max; /* per sample: mean over the XCDs that report a clock */
    double xcd_spread_mhz_max;                                 /* largest max-min over the XCDs within one sample */
    double socket_power_w_mean, socket_power_w_max;            /* current_socket_power samples */
    double energy_j;        /* energy accumulator, last - first sample */
    double energy_power_w;  /* energy_j / seconds (0 when the counter did not move) */
    double ppt_residency_pct, thm_residency_pct; /* accumulated throttler residency over the region, -1 = not reported */
    double gfx_activity_pct_median;
    uint32_t xcds_reporting;
    uint32_t rsmi_index;
    uint64_t bdfid;
    /* round 6: the memory side — is a slow run a lower memory clock, a hotter stack, or the shader clock eating
     * the socket budget?  0 / -1 = the table does not report the field on this firmware. */
    double uclk_mhz_median, uclk_mhz_min;     /* current_uclk: the HBM / memory-controller clock */
    double socclk_mhz_median;                 /* mean of current_socclks[] (data fabric side of the XCDs) */
    double temp_hbm_c_max, temp_mem_c_max, temp_hotspot_c_max; /* hottest HBM stack / memory / hotspot sample */
    double umc_activity_pct_median;           /* average_umc_activity */
    double xcd_busy_spread_pct;               /* per-XCD gfx_busy_acc deltas over the region: (max - min) / max * 100, -1 = not reported */
} gs_summary_t;

static struct {
    int opened;
    uint32_t dv;
    uint64_t bdfid;
    pthread_t thread;
    volatile int running, stop;
    double period_s;
    uint32_t n;
    double t[GS_MAX_SAMPLES];
    float clk_mean[GS_MAX_SAMPLES], clk_spread[GS_MAX_SAMPLES], power[GS_MAX_SAMPLES], activity[GS_MAX_SAMPLES];
    float uclk[GS_MAX_SAMPLES], socclk[GS_MAX_SAMPLES], umc[GS_MAX_SAMPLES];
    float t_hbm_max, t_mem_max, t_hot_max;
    uint32_t xcds;
    rsmi_gpu_metrics_t first, last;
    int have_first;
    char err[256];
} G;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

const char* gs_last_error(void) { return G.err; }

/* "dddd:bb:dd.f" → rocm_smi's BDFID (rocm_smi.h, rsmi_dev_pci_id_get); the partition bits [31:28] are compared masked out */
static int parse_bdf(const char* s, uint64_t* out) {
    unsigned dom = 0, bus = 0, dev = 0, fn = 0;
    if (sscanf(s, "%x:%x:%x.%x", &dom, &bus, &dev, &fn) != 4) return -1;
    *out = ((uint64_t)dom << 32) | ((uint64_t)(bus & 0xff) << 8) | ((uint64_t)(dev & 0x1f) << 3) | (uint64_t)(fn & 0x7);
    return 0;
}

int gs_open(const char* pci_bus_id) {
    uint64_t want = 0;
    G.err[0] = 0;
    if (!pci_bus_id || parse_bdf(pci_bus_id, &want) != 0) {
        snprintf(G.err, sizeof G.err, "unparsable PCI address '%s'", pci_bus_id ? pci_bus_id : "(null)");
        return -1;
    }
    if (!G.opened) {
        rsmi_status_t st = rsmi_init(0);
        if (st != RSMI_STATUS_SUCCESS) {
            snprintf(G.err, sizeof G.err, "rsmi_init failed (%d)", (int)st);
            return -2;
        }
    }
    uint32_t n = 0;
    if (rsmi_num_monitor_devices(&n) != RSMI_STATUS_SUCCESS || n == 0) {
        snprintf(G.err, sizeof G.err, "rocm_smi sees no device");
        return -3;
    }
    const uint64_t mask = ~((uint64_t)0xf << 28);
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t id = 0;
        if (rsmi_dev_pci_id_get(i, &id) != RSMI_STATUS_SUCCESS) continue;
        if ((id & mask) == (want & mask)) {
            G.dv = i;
            G.bdfid = id;
            G.opened = 1;
            return (int)i;
        }
    }
    snprintf(G.err, sizeof G.err, "no rocm_smi device at %s among %u", pci_bus_id, n);
    return -4;
}

static void take_sample(void) {
    rsmi_gpu_metrics_t m;
    memset(&m, 0, sizeof m);
    if (rsmi_dev_gpu_metrics_info_get(G.dv, &m) != RSMI_STATUS_SUCCESS) return;
    if (!G.have_first) {
        G.first = m;
        G.have_first = 1;
    }
    G.last = m;
    if (G.n >= GS_MAX_SAMPLES) return;
    double sum = 0.0, lo = 1e9, hi = 0.0;
    uint32_t cnt = 0;
    for (int x = 0; x < RSMI_MAX_NUM_GFX_CLKS; ++x) {
        uint16_t c = m.current_gfxclks[x];
        if (c == 0 || c == 0xFFFF) continue;
        sum += c;
        if (c < lo) lo = c;
        if (c > hi) hi = c;
        ++cnt;
    }
    if (cnt == 0 && m.current_gfxclk != 0 && m.current_gfxclk != 0xFFFF) {
        sum = lo = hi = m.current_gfxclk;
        cnt = 1;
    }
    if (cnt > G.xcds) G.xcds = cnt;
    uint32_t i = G.n;
    G.t[i] = now_s();
    G.clk_mean[i] = cnt ? (float)(sum / cnt) : 0.f;
    G.clk_spread[i] = cnt ? (float)(hi - lo) : 0.f;
    uint16_t p = m.current_socket_power;
    if (p == 0 || p == 0xFFFF) p = m.average_socket_power;
    G.power[i] = (p == 0xFFFF) ? 0.f : (float)p;
    G.activity[i] = (m.average_gfx_activity == 0xFFFF) ? -1.f : (float)m.average_gfx_activity;
    G.uclk[i] = (m.current_uclk == 0xFFFF) ? 0.f : (float)m.current_uclk;
    {
        double ss = 0.0;
        uint32_t sc = 0;
        for (int x = 0; x < RSMI_MAX_NUM_CLKS; ++x) {
            uint16_t c = m.current_socclks[x];
            if (c == 0 || c == 0xFFFF) continue;
            ss += c;
            ++sc;
        }
        if (sc == 0 && m.current_socclk != 0 && m.current_socclk != 0xFFFF) ss = m.current_socclk, sc = 1;
        G.socclk[i] = sc ? (float)(ss / sc) : 0.f;
    }
    G.umc[i] = (m.average_umc_activity == 0xFFFF) ? -1.f : (float)m.average_umc_activity;
    for (int x = 0; x < RSMI_NUM_HBM_INSTANCES; ++x)
        if (m.temperature_hbm[x] != 0xFFFF && (float)m.temperature_hbm[x] > G.t_hbm_max) G.t_hbm_max = (float)m.temperature_hbm[x];
    if (m.temperature_mem != 0xFFFF && (float)m.temperature_mem > G.t_mem_max) G.t_mem_max = (float)m.temperature_mem;
    if (m.temperature_hotspot != 0xFFFF && (float)m.temperature_hotspot > G.t_hot_max) G.t_hot_max = (float)m.temperature_hotspot;
    G.n = i + 1;
}

static void* loop(void* arg) {
    (void)arg;
    struct timespec ts;
    ts.tv_sec = (time_t)G.period_s;
    ts.tv_nsec = (long)((G.period_s - (double)ts.tv_sec) * 1e9);
    while (!G.stop) {
        take_sample();
        nanosleep(&ts, NULL);
    }
    take_sample();
    return NULL;
}

int gs_start(double period_s) {
    if (!G.opened || G.running) return -1;
    G.n = 0;
    G.t_hbm_max = G.t_mem_max = G.t_hot_max = 0.f;
    G.xcds = 0;
    G.have_first = 0;
    G.stop = 0;
    G.period_s = period_s > 0 ? period_s : 0.002;
    if (pthread_create(&G.thread, NULL, loop, NULL) != 0) return -2;
    G.running = 1;
    return 0;
}

static int cmp_f(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

static double median_f(const float* v, uint32_t n) {
    if (!n) return 0.0;
    float* c = (float*)malloc(n * sizeof(float));
    memcpy(c, v, n * sizeof(float));
    qsort(c, n, sizeof(float), cmp_f);
    double m = (n & 1) ? c[n / 2] : 0.5 * (c[n / 2 - 1] + c[n / 2]);
    free(c);
    return m;
}

int gs_stop(gs_summary_t* out) {
    if (!G.running) return -1;
    G.stop = 1;
    pthread_join(G.thread, NULL);
    G.running = 0;
    if (!out) return 0;
    memset(out, 0, sizeof *out);
    out->rsmi_index = G.dv;
    out->bdfid = G.bdfid;
    out->samples = G.n;
    out->ppt_residency_pct = out->thm_residency_pct = -1.0;
    if (G.n == 0) return 0;
    out->available = 1;
    out->seconds = G.t[G.n - 1] - G.t[0];
    out->xcds_reporting = G.xcds;
    out->gfxclk_mhz_median = median_f(G.clk_mean, G.n);
    out->gfx_activity_pct_median = median_f(G.activity, G.n);
    double lo = 1e9, hi = 0, sp = 0, pw = 0, pmax = 0;
    for (uint32_t i = 0; i < G.n; ++i) {
        if (G.clk_mean[i] < lo) lo = G.clk_mean[i];
        if (G.clk_mean[i] > hi) hi = G.clk_mean[i];
        if (G.clk_spread[i] > sp) sp = G.clk_spread[i];
        pw += G.power[i];
        if (G.power[i] > pmax) pmax = G.power[i];
    }
    out->gfxclk_mhz_min = lo;
    out->gfxclk_mhz_max = hi;
    out->xcd_spread_mhz_max = sp;
    out->socket_power_w_mean = pw / G.n;
    out->socket_power_w_max = pmax;
    /* energy accumulator: 15.259 uJ units (2^-16 J), rocm_smi.h */
    if (G.last.energy_accumulator > G.first.energy_accumulator && G.first.energy_accumulator != 0 &&
        G.last.energy_accumulator != UINT64_MAX) {
        out->energy_j = (double)(G.last.energy_accumulator - G.first.energy_accumulator) * 15.259e-6;
        if (out->seconds > 0) out->energy_power_w = out->energy_j / out->seconds;
    }
    out->uclk_mhz_median = median_f(G.uclk, G.n);
    out->socclk_mhz_median = median_f(G.socclk, G.n);
    out->umc_activity_pct_median = median_f(G.umc, G.n);
    {
        double ulo = 1e9;
        for (uint32_t i = 0; i < G.n; ++i)
            if (G.uclk[i] > 0.f && G.uclk[i] < ulo) ulo = G.uclk[i];
        out->uclk_mhz_min = ulo < 1e9 ? ulo : 0.0;
    }
    out->temp_hbm_c_max = G.t_hbm_max;
    out->temp_mem_c_max = G.t_mem_max;
    out->temp_hotspot_c_max = G.t_hot_max;
    out->xcd_busy_spread_pct = -1.0;
    {   /* per-XCD busy accumulators of partition 0 (the whole GPU in SPX mode): how evenly the launch kept the XCDs busy */
        double bmax = 0.0, bmin = 1e30;
        uint32_t bc = 0;
        for (int x = 0; x < RSMI_MAX_NUM_XCC; ++x) {
            uint64_t b0 = G.first.xcp_stats[0].gfx_busy_acc[x], b1 = G.last.xcp_stats[0].gfx_busy_acc[x];
            if (b0 == UINT64_MAX || b1 == UINT64_MAX || b1 <= b0) continue;
            double d = (double)(b1 - b0);
            if (d > bmax) bmax = d;
            if (d < bmin) bmin = d;
            ++bc;
        }
        if (bc >= 2 && bmax > 0.0) out->xcd_busy_spread_pct = 100.0 * (bmax - bmin) / bmax;
    }
    uint64_t a0 = G.first.accumulation_counter, a1 = G.last.accumulation_counter;
    if (a1 > a0 && a1 != UINT64_MAX && a0 != UINT64_MAX) {
        double d = (double)(a1 - a0);
        if (G.last.ppt_residency_acc != UINT64_MAX && G.last.ppt_residency_acc >= G.first.ppt_residency_acc)
            out->ppt_residency_pct = 100.0 * (double)(G.last.ppt_residency_acc - G.first.ppt_residency_acc) / d;
        if (G.last.socket_thm_residency_acc != UINT64_MAX && G.last.socket_thm_residency_acc >= G.first.socket_thm_residency_acc)
            out->thm_residency_pct = 100.0 * (double)(G.last.socket_thm_residency_acc - G.first.socket_thm_residency_acc) / d;
    }
    return 0;
}

/* one-shot dump for the probe script: every field a reader would want to sanity-check against rocm-smi */
int gs_dump(char* buf, int cap) {
    if (!G.opened) return -1;
    rsmi_gpu_metrics_t m;
    memset(&m, 0, sizeof m);
    rsmi_status_t st = rsmi_dev_gpu_metrics_info_get(G.dv, &m);
    if (st != RSMI_STATUS_SUCCESS) {
        snprintf(buf, cap, "rsmi_dev_gpu_metrics_info_get failed (%d)", (int)st);
        return -2;
    }
    int o = snprintf(buf, cap,
                     "rsmi_index %u bdfid 0x%llx table v%u.%u size %u | current_socket_power %u W average_socket_power %u W | "
                     "current_gfxclk %u average_gfxclk %u | gfx_activity %u%% umc_activity %u%% | energy_acc %llu | "
                     "accumulation_counter %llu ppt_residency_acc %llu socket_thm_residency_acc %llu | throttle 0x%x | gfxclks",
                     G.dv, (unsigned long long)G.bdfid, m.common_header.format_revision, m.common_header.content_revision,
                     m.common_header.structure_size, m.current_socket_power, m.average_socket_power, m.current_gfxclk,
                     m.average_gfxclk_frequency, m.average_gfx_activity, m.average_umc_activity,
                     (unsigned long long)m.energy_accumulator, (unsigned long long)m.accumulation_counter,
                     (unsigned long long)m.ppt_residency_acc, (unsigned long long)m.socket_thm_residency_acc, m.throttle_status);
    for (int x = 0; x < RSMI_MAX_NUM_GFX_CLKS && o < cap - 8; ++x) o += snprintf(buf + o, cap - o, " %u", m.current_gfxclks[x]);
    return 0;
}
