// MultiGpuBatchQPSolver (include/sqp_hip/multi_gpu.hpp): split arithmetic (host only), then the sharded solve with G = 1, 2,
// ... up to every visible device against the single-device BatchQPSolver — bit-identical results in the unsharded order.
// `multi_gpu_test.bin split` runs the host-only part.  Exit 0 = passed, 3 = no HIP device.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sqp_hip/multi_gpu.hpp"

using namespace qp_solver;
#define CHECK(cond)                                                                 \
    do {                                                                            \
        if (!(cond)) {                                                              \
            fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

static void split_arithmetic() {
    for (long long total : {1LL, 2LL, 7LL, 8LL, 9LL, 1023LL, 65536LL, 65537LL}) {
        for (int parts : {1, 2, 3, 4, 7, 8}) {
            long long prev = 0, smallest = total, largest = 0;
            for (int g = 0; g < parts; g++) {
                long long lo, hi;
                sqph_shard_bounds(total, parts, g, &lo, &hi);
                CHECK(lo == prev && hi >= lo);  // contiguous, in order
                prev = hi;
                smallest = hi - lo < smallest ? hi - lo : smallest;
                largest = hi - lo > largest ? hi - lo : largest;
            }
            CHECK(prev == total);               // covers the batch exactly
            CHECK(largest - smallest <= 1);     // balanced
        }
    }
    long long lo, hi;
    sqph_shard_bounds(65536, 8, 3, &lo, &hi);  // BASELINE configs[2]: 8,192 per GPU
    CHECK(lo == 3 * 8192 && hi == 4 * 8192);
}

static int g_cross_device_rccl_runs = 0;  // sharded solves whose gather crossed devices over RCCL (0 on a one-GPU box)

struct Lcg {
    unsigned long long s;
    double uni() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; }
};

static void sharded_vs_single(int n, int m, int B) {
    Lcg g{99};
    std::vector<double> P((size_t)B * n * n), q((size_t)B * n), A((size_t)B * m * n), l((size_t)B * m), u((size_t)B * m);
    for (int b = 0; b < B; b++) {
        double *Pb = &P[(size_t)b * n * n];
        std::vector<double> G((size_t)n * n);
        for (auto &v : G) v = g.uni() - 0.5;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                double s = 0;
                for (int k = 0; k < n; k++) s += G[(size_t)k * n + i] * G[(size_t)k * n + j];
                Pb[(size_t)j * n + i] = s / n + (i == j ? 0.1 : 0.0);
            }
        for (int j = 0; j < n; j++) q[(size_t)b * n + j] = g.uni() - 0.5;
        for (int e = 0; e < m * n; e++) A[(size_t)b * m * n + e] = g.uni() - 0.5;
        for (int i = 0; i < m; i++) { l[(size_t)b * m + i] = -g.uni(); u[(size_t)b * m + i] = g.uni(); }
    }
    BatchQPSolver<double> single(n, m, B);
    single.settings().max_iter = 60;
    single.settings().check_termination = 0;
    single.setup_solve(single.packed(B, P.data(), q.data(), A.data(), l.data(), u.data()));
    const int ndev = sqph_device_count();
    // every device count available, then shard counts beyond it placed round-robin (several shards per GPU: the G > 1 split,
    // per-shard streams and the gather offsets run even where one GPU is all there is)
    std::vector<std::vector<int>> placements;
    for (int G = 1; G <= ndev; G++) {
        std::vector<int> d;
        for (int g = 0; g < G; g++) d.push_back(g);
        placements.push_back(d);
    }
    for (int G : {2, 3, 5, 8}) {
        std::vector<int> d;
        for (int g = 0; g < G; g++) d.push_back(g % ndev);
        placements.push_back(d);
    }
    // every placement twice: default transport (RCCL between different devices, a device-to-device copy on the root's own), and
    // with every shard's records forced through RCCL (a send to self on a one-GPU box: librccl opened, communicators created,
    // grouped ncclSend / ncclRecv on the producer's and the gather's streams)
    for (int pass = 0; pass < 2; pass++)
    for (const auto &devices : placements) {
        MultiGpuBatchQPSolver<double> multi(n, m, B, devices, 0, 0, pass ? SQPH_GATHER_RCCL_ALWAYS : 0);
        const int G = (int)devices.size();
        CHECK(multi.num_devices() == (G < B ? G : B));
        multi.settings() = single.settings();
        multi.setup_solve(multi.packed(P.data(), q.data(), A.data(), l.data(), u.data()));
        for (int b = 0; b < B; b++) {
            CHECK(!std::memcmp(multi.primal_solution(b), single.primal_solution(b), sizeof(double) * n));
            CHECK(!std::memcmp(multi.dual_solution(b), single.dual_solution(b), sizeof(double) * m));
            CHECK(multi.info(b).iter == single.info(b).iter && multi.info(b).status == single.info(b).status);
        }
        void *dx = nullptr, *dy = nullptr;
        sqph_info *di = nullptr;
        multi.gathered_device(&dx, &dy, &di);
        CHECK(dx && dy && di);
        // a second call on the same object (solve() on the resident factors of every shard) gathers again
        multi.solve(multi.packed(P.data(), q.data(), A.data(), l.data(), u.data()));
        single.solve(single.packed(B, P.data(), q.data(), A.data(), l.data(), u.data()));
        for (int b = 0; b < B; b++) CHECK(!std::memcmp(multi.primal_solution(b), single.primal_solution(b), sizeof(double) * n));
        single.setup_solve(single.packed(B, P.data(), q.data(), A.data(), l.data(), u.data()));
        {   // device-resident shards: one launch per device back to back from this thread, every shard's records posted in ONE
            // sqph_gather_post_many (one RCCL group, the root's receives on one stream per source device)
            static void *hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
            CHECK(hip);
            auto hipSetDevice_ = (int (*)(int))dlsym(hip, "hipSetDevice");
            auto hipMalloc_ = (int (*)(void **, size_t))dlsym(hip, "hipMalloc");
            auto hipMemcpy_ = (int (*)(void *, const void *, size_t, int))dlsym(hip, "hipMemcpy");
            auto hipFree_ = (int (*)(void *))dlsym(hip, "hipFree");
            CHECK(hipSetDevice_ && hipMalloc_ && hipMemcpy_ && hipFree_);
            std::vector<MultiGpuBatchQPSolver<double>::Batch> shards;
            std::vector<void *> bufs;
            for (int gi = 0; gi < multi.num_devices(); gi++) {
                const long long lo = multi.shard_begin(gi), hi = multi.shard_end(gi);
                const size_t c = (size_t)(hi - lo);
                CHECK(hipSetDevice_(devices[(size_t)gi]) == 0);
                const double *src[5] = {&P[(size_t)lo * n * n], &q[(size_t)lo * n], &A[(size_t)lo * m * n], &l[(size_t)lo * m], &u[(size_t)lo * m]};
                const size_t len[5] = {c * n * n, c * n, c * (size_t)m * n, c * m, c * m};
                double *d[5];
                for (int k = 0; k < 5; k++) {
                    CHECK(hipMalloc_((void **)&d[k], len[k] * sizeof(double)) == 0);
                    CHECK(hipMemcpy_(d[k], src[k], len[k] * sizeof(double), 1 /* hipMemcpyHostToDevice */) == 0);
                    bufs.push_back(d[k]);
                }
                MultiGpuBatchQPSolver<double>::Batch b{(int)c, SQPH_DEVICE, d[0], d[1], d[2], d[3], d[4], (long long)n * n, n, (long long)m * n, m, m};
                shards.push_back(b);
            }
            CHECK(hipSetDevice_(0) == 0);
            multi.setup_solve_device(shards);
            for (int b = 0; b < B; b++) {
                CHECK(!std::memcmp(multi.primal_solution(b), single.primal_solution(b), sizeof(double) * n));
                CHECK(!std::memcmp(multi.dual_solution(b), single.dual_solution(b), sizeof(double) * m));
                CHECK(multi.info(b).iter == single.info(b).iter && multi.info(b).status == single.info(b).status);
            }
            for (void *bp : bufs) hipFree_(bp);
        }
        bool distinct = true;
        for (size_t i = 0; i < devices.size(); i++) distinct = distinct && devices[i] == (int)i;
        if (pass || (distinct && G > 1)) CHECK(!std::strcmp(multi.gather_transport(), "rccl"));
        // the first box with two devices tests the real path without edits: placements {0, 1, ...} (distinct devices) MUST have moved
        // their records over RCCL between devices — asserted above — with the bit-identity checks of this loop behind them
        if (distinct && G > 1 && !std::strcmp(multi.gather_transport(), "rccl")) g_cross_device_rccl_runs++;
        printf("multi-GPU n=%d m=%d batch=%d over %d shard(s) on %d device(s), gather transport %s: gathered records bit-identical to the single-device solve\n", n, m, B,
               multi.num_devices(), ndev < G ? ndev : G, multi.gather_transport());
    }
}

// host-memspace round trip of a C3-shaped batch (PCIe-inclusive): one handle against several shards on the same GPU, whose
// host-to-device copies (one host thread each) overlap each other and the kernels of the shards already on the device
#include <chrono>
static void pcie_shards(int n, int m, int B) {
    std::vector<double> P((size_t)B * n * n, 0.0), q((size_t)B * n), A((size_t)B * m * n), l((size_t)B * m), u((size_t)B * m);
    Lcg g{99};
    for (int b = 0; b < B; b++) {
        for (int i = 0; i < n; i++) P[(size_t)b * n * n + (size_t)i * n + i] = 1.0 + g.uni();
        for (int j = 0; j < n; j++) q[(size_t)b * n + j] = g.uni() - 0.5;
        for (int e = 0; e < m * n; e++) A[(size_t)b * m * n + e] = g.uni() - 0.5;
        for (int i = 0; i < m; i++) { l[(size_t)b * m + i] = -g.uni(); u[(size_t)b * m + i] = g.uni(); }
    }
    for (int G : {1, 2, 4, 8}) {
        std::vector<int> d((size_t)G, 0);
        MultiGpuBatchQPSolver<double> multi(n, m, B, d);
        multi.settings().max_iter = 200;
        multi.settings().check_termination = 0;
        auto b = multi.packed(P.data(), q.data(), A.data(), l.data(), u.data());
        for (int w = 0; w < 2; w++) { multi.setup_solve(b); (void)multi.primal_solution(0); }
        const int K = 5;
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < K; k++) { multi.setup_solve(b); (void)multi.primal_solution(0); }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / K;
        printf("pcie n=%d m=%d batch=%d host-memspace round trip, %d shard(s) on one GPU: %.2f ms = %.3g QP/s\n", n, m, B, G, ms, B / (ms * 1e-3));
    }
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "pcie")) {
        pcie_shards(50, 100, 8192);
        return 0;
    }
    split_arithmetic();
    if (argc > 1 && !strcmp(argv[1], "split")) {
        printf("split arithmetic passed\n");
        return 0;
    }
    try {
        sharded_vs_single(20, 40, 37);
        sharded_vs_single(50, 100, 64);
    } catch (const std::runtime_error &e) {
        if (std::string(e.what()).find("no HIP device") != std::string::npos) {
            fprintf(stderr, "no HIP device: %s\n", e.what());
            return 3;
        }
        fprintf(stderr, "exception: %s\n", e.what());
        return 2;
    }
    // with >= 2 devices the cross-device RCCL gather MUST have run (both passes x both shapes x every G in 2..ndev)
    if (sqph_device_count() >= 2) CHECK(g_cross_device_rccl_runs >= 4 * (sqph_device_count() - 1));
    printf("multi_gpu_test: all passed (%d device(s); %d sharded solves gathered across devices over RCCL)\n", sqph_device_count(), g_cross_device_rccl_runs);
    return 0;
}
