"""Diagnostic: board power and shader clock while a command runs (sysfs hwmon of the first amdgpu card, 20 ms samples).
    python tools/power_probe.py -- python bench.py --steps 300 --no-cpu-baseline --no-secondary --no-kernel-events
Prints min / median / max of the samples taken in the middle 60 % of the run, the power cap, and the busy fraction."""
import glob, os, subprocess, sys, time, statistics


def read(p):
    try:
        return open(p).read().strip()
    except OSError:
        return None


def cards():
    return [hw for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")) if read(hw + "/name") == "amdgpu"]


def sample(hw):
    dev = os.path.dirname(os.path.dirname(hw))
    pw = read(hw + "/power1_average") or read(hw + "/power1_input")
    fq, mq, busy = read(hw + "/freq1_input"), read(hw + "/freq2_input"), read(dev + "/gpu_busy_percent")
    return (float(pw) / 1e6 if pw else None, float(fq) / 1e6 if fq else None, float(mq) / 1e6 if mq else None,
            float(busy) if busy else None)


hws = cards()                          # the box shows every card of the host; the one the job runs on is the one that gets busy
cmd = sys.argv[sys.argv.index("--") + 1:]
print(len(hws), "amdgpu hwmon nodes")
p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
per = {hw: [] for hw in hws}
t0 = time.time()
while p.poll() is None:
    t = time.time() - t0
    for hw in hws:
        per[hw].append((t,) + sample(hw))
    time.sleep(0.02)
out = p.stdout.read()
print(out.strip().splitlines()[-1][:300] if out.strip() else "(no output)")


def mean_power(rs):
    v = [r[1] for r in rs[len(rs) // 2:] if r[1] is not None]
    return sum(v) / len(v) if v else 0.0


for hw in hws:
    print(f"  {hw.split('/')[4]}: mean power in the second half of the run {mean_power(per[hw]):7.1f} W")
hw = max(hws, key=lambda h: mean_power(per[h]))
rows = per[hw]
cap = read(hw + "/power1_cap")
print("busiest:", hw, "| power cap [W]:", float(cap) / 1e6 if cap else None, "| samples:", len(rows), "| run", round(time.time() - t0, 1), "s")
n = len(rows)
mid = rows[int(0.5 * n):int(0.95 * n)]
for name, i in (("power W", 1), ("sclk MHz", 2), ("mclk MHz", 3), ("busy %", 4)):
    v = [r[i] for r in mid if r[i] is not None]
    if v:
        print(f"{name:9s} min {min(v):8.1f}  median {statistics.median(v):8.1f}  max {max(v):8.1f}   ({len(v)} samples in the timed part)")
    else:
        print(f"{name:9s} not readable")
