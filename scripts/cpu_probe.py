import os, time, subprocess, sys
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpuset/cpuset.cpus"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, "-")
code = "import time\nt=time.perf_counter()\nx=0\nfor i in range(6000000): x+=i*i\nprint(time.perf_counter()-t)"
for n in (1, 8, 16, 32, 64, 128):
    t0 = time.perf_counter()
    ps = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True) for _ in range(n)]
    ts = [float(p.communicate()[0]) for p in ps]
    print(n, "procs: wall %.2f s, per-proc mean %.2f s" % (time.perf_counter() - t0, sum(ts) / len(ts)), flush=True)
