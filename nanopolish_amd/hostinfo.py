"""Host facts the measurement scripts need (not on the product path)."""
import os


def usable_cores():
    """Hardware threads this process may actually run on: the affinity mask, capped by the cgroup CPU quota (a container that
    SEES 256 cores but is throttled to a fraction of them is the usual reason an OpenMP run with 256 threads crawls)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]               # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    eff = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return n, quota, eff
