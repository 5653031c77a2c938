// Is another PROCESS computing on this GPU?  The persistent kernels (fused residual stack, tail) assume that all their
// workgroups are resident at once - true while the device is the engine's own.  The kernel driver publishes who else is
// there: /sys/class/kfd/kfd/proc/<pid>/queues/<n>/gpuid lists every process's compute queues per GPU, and
// .../stats_<gpuid>/cu_occupancy the CUs that process's waves hold right now.  The pids are the HOST's (a container cannot
// find its own entry by getpid()), so the question is asked without knowing which entry is ours:
//   holders  = processes with at least one queue on our GPU                      (we are one of them once HIP is up)
//   busy_cus = the sum of their cu_occupancy, confirmed after OUR engine's stream has drained (so what is busy is somebody
//              else's - another process, or another stream of this one: either way not the engine's own device)
// holders >= 2 and busy_cus > 0  =>  the engine yields: one launch per phase from then on (abi.hip: yield_fused).
// Idle co-holders (a sibling rank's context, a launcher, a notebook that imported torch) do not count, and a tenant that
// arrives between the scan and the launch is what the barriers' ~1 s spin bound remains the backstop for.
// Where the files are not readable (no sysfs in the container, another driver) the answer is "unknown": exclusive is assumed.
// Host code; plain POSIX.
#pragma once
#include <dirent.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace drh {

struct TenantScan {
    int holders = 0;       // processes with >= 1 queue on the GPU
    long busy_cus = 0;     // sum of their cu_occupancy
    bool readable = false; // the proc directory could be listed at all
};

inline bool read_long(const std::string& path, long* out) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[64];
    const bool ok = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!ok) return false;
    char* end = nullptr;
    const long v = strtol(buf, &end, 10);
    if (end == buf) return false;
    *out = v;
    return true;
}

// KFD gpu_id of the PCI function (domain, bus, device, function 0): topology/nodes/<n>/{gpu_id, properties}
inline long kfd_gpu_id(const std::string& root, int domain, int bus, int device) {
    const std::string nodes = root + "/topology/nodes";
    DIR* d = opendir(nodes.c_str());
    if (!d) return -1;
    long found = -1;
    const long want_loc = ((long)bus << 8) | ((long)device << 3);
    while (dirent* ent = readdir(d)) {
        if (ent->d_name[0] == '.') continue;
        const std::string nd = nodes + "/" + ent->d_name;
        long gid = 0;
        if (!read_long(nd + "/gpu_id", &gid) || gid == 0) continue;
        FILE* f = fopen((nd + "/properties").c_str(), "r");
        if (!f) continue;
        char key[64];
        long val, loc = -1, dom = 0;
        while (fscanf(f, "%63s %ld", key, &val) == 2) {
            if (!strcmp(key, "location_id")) loc = val;
            else if (!strcmp(key, "domain")) dom = val;
        }
        fclose(f);
        if (loc == want_loc && dom == domain) { found = gid; break; }
    }
    closedir(d);
    return found;
}

inline TenantScan scan_tenants(const std::string& root, long gpu_id) {
    TenantScan r;
    const std::string proc = root + "/proc";
    DIR* d = opendir(proc.c_str());
    if (!d) return r;
    r.readable = true;
    while (dirent* ent = readdir(d)) {
        if (ent->d_name[0] == '.') continue;
        const std::string pd = proc + "/" + ent->d_name;
        // (a process that never opened this GPU has no stats_<gpu_id> directory: one syscall rules it out)
        if (access((pd + "/stats_" + std::to_string(gpu_id)).c_str(), F_OK) != 0) continue;
        DIR* q = opendir((pd + "/queues").c_str());
        if (!q) continue;
        bool holds = false;
        while (dirent* qe = readdir(q)) {
            if (qe->d_name[0] == '.') continue;
            long g = -1;
            if (read_long(pd + "/queues/" + qe->d_name + "/gpuid", &g) && g == gpu_id) { holds = true; break; }
        }
        closedir(q);
        if (!holds) continue;
        r.holders += 1;
        long cu = 0;
        if (read_long(pd + "/stats_" + std::to_string(gpu_id) + "/cu_occupancy", &cu) && cu > 0) r.busy_cus += cu;
    }
    closedir(d);
    return r;
}

}  // namespace drh
