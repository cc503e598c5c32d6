// smi_sampler.cpp — a SEPARATE process that samples socket power, clocks and throttle residencies of every GPU the SMI library
// sees, every `period_ms` (default 10), into a CSV, until its stop file appears.  It never touches HIP: it is started before the
// workload initialises the GPU and reads the firmware's gpu_metrics table through librocm_smi64 only.
//
// CSV columns: t_mono_s (CLOCK_MONOTONIC, the clock the workload's case markers use), dev, bdfid, socket_w (current_socket_power, or
// average_socket_power where the table has no current one), gfxclk_mhz (mean of the XCDs' current clocks), gfxclk_min, gfxclk_max,
// uclk_mhz (memory clock), hotspot_c, mem_c, gfx_act, umc_act, energy_acc (15.259 uJ units), ppt_acc, thm_acc, hbm_thm_acc, prochot_acc,
// accum_counter (the residency counters' time base: residency fraction = d(ppt_acc) / d(accum_counter)), throttle_status,
// indep_throttle_status, power_cap_w.
//
// build: g++ -O2 -o smi_sampler smi_sampler.cpp -I/opt/rocm/include -L/opt/rocm/lib -lrocm_smi64 -Wl,-rpath,/opt/rocm/lib
// run:   ./smi_sampler out.csv stopfile [period_ms]
#include <rocm_smi/rocm_smi.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <sys/stat.h>

static double mono() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: smi_sampler out.csv stopfile [period_ms]\n"); return 2; }
  const double period = (argc > 3 ? atof(argv[3]) : 10.0) * 1e-3;
  rsmi_status_t rc = rsmi_init(0);
  if (rc != RSMI_STATUS_SUCCESS) { fprintf(stderr, "rsmi_init: %d\n", (int)rc); return 1; }
  uint32_t ndev = 0;
  rsmi_num_monitor_devices(&ndev);
  FILE *f = fopen(argv[1], "w");
  if (!f) { perror(argv[1]); return 1; }
  fprintf(f, "t_mono_s,dev,bdfid,socket_w,gfxclk_mhz,gfxclk_min,gfxclk_max,uclk_mhz,hotspot_c,mem_c,gfx_act,umc_act,energy_acc,ppt_acc,thm_acc,"
             "hbm_thm_acc,prochot_acc,accum_counter,throttle_status,indep_throttle_status,power_cap_w\n");
  uint64_t bdf[64] = {0}, cap[64] = {0};
  if (ndev > 64) ndev = 64;
  for (uint32_t d = 0; d < ndev; d++) {
    rsmi_dev_pci_id_get(d, &bdf[d]);
    rsmi_dev_power_cap_get(d, 0, &cap[d]);            // microwatts
  }
  fprintf(stderr, "smi_sampler: %u devices, period %.1f ms\n", ndev, period * 1e3);
  struct stat sb;
  uint64_t nsamp = 0;
  double next = mono();
  while (stat(argv[2], &sb) != 0) {
    for (uint32_t d = 0; d < ndev; d++) {
      rsmi_gpu_metrics_t m;
      memset(&m, 0xFF, sizeof m);
      const double t = mono();
      if (rsmi_dev_gpu_metrics_info_get(d, &m) != RSMI_STATUS_SUCCESS) continue;
      double sum = 0; int cnt = 0; uint32_t mn = 0xFFFF, mx = 0;
      for (int i = 0; i < RSMI_MAX_NUM_GFX_CLKS; i++) {
        const uint16_t c = m.current_gfxclks[i];
        if (c == 0xFFFF || c == 0) continue;
        sum += c; cnt++; if (c < mn) mn = c; if (c > mx) mx = c;
      }
      if (!cnt && m.current_gfxclk != 0xFFFF) { sum = mn = mx = m.current_gfxclk; cnt = 1; }
      const unsigned w = m.current_socket_power != 0xFFFF ? m.current_socket_power : m.average_socket_power;
      fprintf(f, "%.6f,%u,%llu,%u,%.1f,%u,%u,%u,%u,%u,%u,%u,%llu,%llu,%llu,%llu,%llu,%llu,%u,%llu,%.0f\n", t, d, (unsigned long long)bdf[d], w,
              cnt ? sum / cnt : 0.0, cnt ? mn : 0, mx, (unsigned)m.current_uclk, (unsigned)m.temperature_hotspot, (unsigned)m.temperature_mem,
              (unsigned)m.average_gfx_activity, (unsigned)m.average_umc_activity, (unsigned long long)m.energy_accumulator,
              (unsigned long long)m.ppt_residency_acc, (unsigned long long)m.socket_thm_residency_acc, (unsigned long long)m.hbm_thm_residency_acc,
              (unsigned long long)m.prochot_residency_acc, (unsigned long long)m.accumulation_counter, (unsigned)m.throttle_status,
              (unsigned long long)m.indep_throttle_status, cap[d] / 1e6);
      nsamp++;
    }
    if ((nsamp & 0x3FF) == 0) fflush(f);
    next += period;
    const double now = mono();
    if (next > now) usleep((useconds_t)((next - now) * 1e6)); else next = now;
  }
  fclose(f);
  rsmi_shut_down();
  fprintf(stderr, "smi_sampler: %llu samples\n", (unsigned long long)nsamp);
  return 0;
}
