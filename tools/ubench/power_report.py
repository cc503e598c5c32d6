#!/usr/bin/env python3
"""power_report.py — cut tools/ubench/smi_sampler's CSV by the case windows of power_cases.py: watts, clocks, throttle residency per case.

  python tools/ubench/power_report.py samples.csv cases.json out.json

Per case (the first 0.5 s of a window is left out: ramp): socket power (mean, 5th / 95th percentile; and from the firmware's energy
accumulator over the window, 15.259 uJ units), mean shader clock over the XCDs and its min / max, memory clock, hot-spot and HBM
temperature, the fraction of the window the firmware spent limiting for POWER (d ppt_residency_acc / d accumulation_counter), for
temperature, for HBM temperature and under PROCHOT, and the sampling period actually achieved."""
import csv, json, sys
import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
cases = json.load(open(sys.argv[2]))
devs = sorted({int(r["dev"]) for r in rows})


def bdfid_of(s):                     # "0000:0c:00.0" -> the SMI's 64-bit id
    dom, bus, rest = s.split(":")
    d, fn = rest.split(".")
    return (int(dom, 16) << 32) | (int(bus, 16) << 8) | (int(d, 16) << 3) | int(fn, 16)


pick = None
if cases.get("bdf"):
    want = bdfid_of(cases["bdf"])
    for r in rows:
        if int(r["bdfid"]) & 0xFFFFFFFFFFFF == want & 0xFFFFFFFFFFFF or (int(r["bdfid"]) & 0xFFFF) == (want & 0xFFFF):
            pick = int(r["dev"]); break
if pick is None:                     # the device whose power moves most
    spread = {d: np.ptp([float(r["socket_w"]) for r in rows if int(r["dev"]) == d]) for d in devs}
    pick = max(spread, key=spread.get)
R = [r for r in rows if int(r["dev"]) == pick]
t = np.array([float(r["t_mono_s"]) for r in R])
col = lambda k: np.array([float(r[k]) for r in R])
W, clk, cmin, cmax, uclk, hot, mem = col("socket_w"), col("gfxclk_mhz"), col("gfxclk_min"), col("gfxclk_max"), col("uclk_mhz"), col("hotspot_c"), col("mem_c")
en, ppt, thm, hbm, pro, acc = col("energy_acc"), col("ppt_acc"), col("thm_acc"), col("hbm_thm_acc"), col("prochot_acc"), col("accum_counter")
out = dict(device=cases["device"], bdf=cases.get("bdf"), smi_device_index=pick, smi_devices_seen=len(devs), power_cap_w=float(R[0]["power_cap_w"]),
           samples=len(R), sampling_period_ms=round(float(np.median(np.diff(t))) * 1e3, 2), sampler_started_before_gpu_touch_s=round(cases["t_gpu_first_touch"] - t[0], 2),
           n=cases["n"], placement=cases["placement"], key_window=cases["key_window"], cases=[])
for c in cases["cases"]:
    sel = (t >= c["t0"] + 0.5) & (t <= c["t1"])
    if sel.sum() < 3:
        continue
    i0, i1 = np.flatnonzero(sel)[[0, -1]]
    dacc = acc[i1] - acc[i0]
    frac = lambda x: round(float((x[i1] - x[i0]) / dacc), 4) if dacc > 0 and x[i1] < 1e18 else None
    e = dict(case=c["name"], seconds=round(c["t1"] - c["t0"], 2), samples=int(sel.sum()),
             socket_w_mean=round(float(W[sel].mean()), 1), socket_w_p5=float(np.percentile(W[sel], 5)), socket_w_p95=float(np.percentile(W[sel], 95)),
             socket_w_from_energy=round(float((en[i1] - en[i0]) * 15.259e-6 / (t[i1] - t[i0])), 1) if en[i1] < 1e19 else None,
             gfxclk_mhz_mean=round(float(clk[sel].mean()), 1), gfxclk_mhz_min=float(cmin[sel].min()), gfxclk_mhz_max=float(cmax[sel].max()),
             uclk_mhz_mean=round(float(uclk[sel].mean()), 1), hotspot_c_max=float(hot[sel].max()), hbm_c_max=float(mem[sel].max()),
             ppt_limited_fraction=frac(ppt), thermal_limited_fraction=frac(thm), hbm_thermal_limited_fraction=frac(hbm), prochot_fraction=frac(pro),
             throttle_status_seen=sorted({int(float(r["throttle_status"])) for r, s in zip(R, sel) if s})[:8])
    for k in ("ms_per_iter", "m_steps_per_s", "iters", "g_field_mul_per_s"):
        if k in c:
            e[k] = round(c[k], 4) if isinstance(c[k], float) else c[k]
    out["cases"].append(e)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(f"device {out['device']} (SMI index {pick} of {len(devs)}), cap {out['power_cap_w']:.0f} W, sampling every {out['sampling_period_ms']} ms")
print(f"{'case':12s} {'W mean':>7s} {'p5':>6s} {'p95':>6s} {'W(energy)':>9s} {'sclk':>7s} {'min':>6s} {'max':>6s} {'mclk':>6s} {'ppt':>6s} {'thm':>6s} {'hot C':>6s}  rate")
for e in out["cases"]:
    print(f"{e['case']:12s} {e['socket_w_mean']:7.1f} {e['socket_w_p5']:6.0f} {e['socket_w_p95']:6.0f} {str(e['socket_w_from_energy']):>9s} {e['gfxclk_mhz_mean']:7.1f} "
          f"{e['gfxclk_mhz_min']:6.0f} {e['gfxclk_mhz_max']:6.0f} {e['uclk_mhz_mean']:6.0f} {str(e['ppt_limited_fraction']):>6s} {str(e['thermal_limited_fraction']):>6s} "
          f"{e['hotspot_c_max']:6.0f}  {e.get('m_steps_per_s', e.get('g_field_mul_per_s', ''))}")
