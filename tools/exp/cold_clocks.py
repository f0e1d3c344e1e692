#!/usr/bin/env python3
"""(GPU) What does the FIRST scan of a batch cost beyond a warm one, and is it the GPU's clocks?  The shard is scanned in short
series - back to back, after an idle gap, right behind the input generator - with the sample of the adaptive width taken out
(FRISK_SCAN_BITS4 forces the plain 4-bit bulk) and with it in; a side thread samples the shader clock (sysfs pp_dpm_sclk /
rocm-smi) every millisecond or so.  One JSON line per series: HIP-event time of each scan and the clock readings around it."""
import glob, json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd import Engine, synth

def sclk_reader():
    paths = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
    def read():
        out = []
        for p in paths:
            try:
                for ln in open(p):
                    if ln.strip().endswith("*"):
                        out.append(ln.split(":")[1].strip().rstrip("*").strip())
            except OSError:
                pass
        return out
    return read if paths and read() else None

class Sampler(threading.Thread):
    def __init__(self, read):
        super().__init__(daemon=True); self.read, self.on, self.log = read, True, []
    def run(self):
        while self.on:
            self.log.append((time.perf_counter(), self.read())); time.sleep(0.0005)

read = sclk_reader()
smi = None
if read is None:
    try:
        smi = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout[-800:]
    except Exception as err:
        smi = "rocm-smi failed: %r" % (err,)
print(json.dumps({"sclk_source": "sysfs pp_dpm_sclk" if read else "none readable", "rocm_smi_tail": smi}), flush=True)
lens = synth.c5_shard_lens(8, 0)
kw = dict(seed=0xC5, island_frac=0.02, n_frac=0.07, lower_frac=0.0)
with Engine(1, 8) as e:
    e.synth(lens, **kw)
    e.profile_reset(); e.profile_add(); e.profile_finalize()
    for _ in range(4):
        e.scan(5000, 1000, pinned=True)
    def series(label, prep, forced, n=4):
        s = Sampler(read) if read else None
        if s: s.start()
        prep()
        t0 = time.perf_counter()
        ms = []
        for _ in range(n):
            e.scan(5000, 1000, pinned=True, bits4=forced); ms.append(round(e.kernel_ms(0), 3))
        t1 = time.perf_counter()
        clk = None
        if s:
            s.on = False; s.join()
            clk = [c for t, c in s.log if t0 - 0.002 <= t <= t1][::max(1, len(s.log) // 40)][:40]
        print(json.dumps({"series": label, "forced_plain_4bit": forced, "scan_kernel_ms": ms, "sclk_during": clk}), flush=True)
    def regen():
        e.synth(lens, **kw); e.profile_reset(); e.profile_add(); e.profile_finalize()
    for forced in (True, False):
        series("back to back", lambda: None, forced)
        series("after 20 ms of idle", lambda: time.sleep(0.02), forced)
        series("after 500 ms of idle", lambda: time.sleep(0.5), forced)
        series("fresh batch (generator + profile, then at once)", regen, forced)
        series("fresh batch, then 20 ms of idle", lambda: (regen(), time.sleep(0.02)), forced)
