// Team size of the host mirror's parallel loops.  OpenMP's default is every hardware thread the machine shows (256 on the GPU box),
// which is not what a process may use there: a container's CPU share is smaller, and a team larger than it turns every barrier
// into a wait for descheduled threads (a 1080p wavefront took 2.5 s instead of 60 ms).  CGRT_HOST_THREADS overrides; otherwise
// min(omp_get_max_threads(), the cgroup's CPU quota if /sys/fs/cgroup/cpu.max states one, 32).
#pragma once
#include <omp.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace cgrt {
inline int hostTeam() {
    static const int team = [] {
        if (const char* e = std::getenv("CGRT_HOST_THREADS")) {
            const int v = std::atoi(e);
            if (v > 0) return v;
        }
        int cap = 32;
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" in microseconds, or "max <period>"
            long long quota = 0, period = 0;
            if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
                cap = std::min(cap, (int)((quota + period - 1) / period));
            std::fclose(f);
        }
        return std::max(1, std::min(omp_get_max_threads(), cap));
    }();
    return team;
}
}  // namespace cgrt
