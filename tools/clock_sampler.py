#!/usr/bin/env python3
"""Sample GPU shader clock / power from sysfs while a command runs (no GPU API calls here).
   python tools/clock_sampler.py OUT.json -- python bench.py ...
Reports the distribution of freq1_input (sclk) and power1_average over the samples taken
while the child was alive; used to state the clock-limited MFMA peak in DESIGN.md."""
import glob, json, subprocess, sys, time

def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None

def main():
    out = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    freq = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
    power = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")) or \
        sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
    dpm = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
    child = subprocess.Popen(cmd)
    samples = []
    t0 = time.time()
    while child.poll() is None:
        row = {"t": round(time.time() - t0, 3)}
        for i, p in enumerate(freq):
            v = read(p)
            if v and v.isdigit():
                row[f"sclk{i}_mhz"] = int(v) / 1e6
        for i, p in enumerate(power):
            v = read(p)
            if v and v.isdigit():
                row[f"power{i}_w"] = int(v) / 1e6
        if not freq:
            for i, p in enumerate(dpm):
                v = read(p) or ""
                cur = [ln for ln in v.splitlines() if ln.endswith("*")]
                if cur:
                    row[f"dpm{i}"] = cur[0]
        samples.append(row)
        time.sleep(0.02)
    keys = sorted({k for r in samples for k in r if k != "t"})
    summary = {"cmd": cmd, "rc": child.returncode, "n_samples": len(samples), "sources": freq + power + dpm}
    for k in keys:
        vals = [r[k] for r in samples if k in r and isinstance(r[k], (int, float))]
        if vals:
            vals.sort()
            summary[k] = {"min": vals[0], "p10": vals[len(vals) // 10], "median": vals[len(vals) // 2],
                          "p90": vals[(9 * len(vals)) // 10], "max": vals[-1]}
    with open(out, "w") as f:
        json.dump({"summary": summary, "samples": samples[::5]}, f)
    print(json.dumps(summary))
    return child.returncode

if __name__ == "__main__":
    sys.exit(main())
