"""Run a command while polling rocm-smi; print the command's output and the mean socket power of the middle of the run.
   python tools/power_run.py tools/ubench/power_modes.bin 0 5"""
import re, subprocess, sys, threading, time
samples, stop = [], False


def poll():
    while not stop:
        o = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True).stdout
        p = re.search(r"Package Power \(W\): ([\d.]+)", o)
        samples.append((time.time(), float(p.group(1)) if p else -1))


th = threading.Thread(target=poll); th.start()
time.sleep(0.5)
idle = samples[-1][1] if samples else -1
t0 = time.time()
out = subprocess.run(sys.argv[1:], capture_output=True, text=True)
t1 = time.time()
stop = True; th.join()
mid = [p for t, p in samples if t0 + 0.35 * (t1 - t0) < t < t1 - 0.1 * (t1 - t0)]
print(out.stdout.strip(), "| power before %.0f W, during (mean of %d samples) %.0f W, max %.0f W" % (idle, len(mid), sum(mid) / max(len(mid), 1), max(mid) if mid else -1))
