// multi_gpu.hpp — the batch behind the C++ API sharded over the GPUs of one node (BASELINE configs[2]).
//
// One process, one libsqp_hip handle per device, each on its own non-blocking stream.  The batch is split into contiguous
// blocks (sqph_shard_bounds); every device runs the fused setup+solve of its block with no data-path exchange; the only
// communication is the collection of the result records (x, y, info) on a root device by peer-to-peer copies over xGMI,
// posted on each producer's stream right behind its solve (sqph_gather_post).  Host-memspace problem data is handed to the
// devices from one host thread per device (each H2D goes over that GPU's own PCIe link); device-memspace shards are
// launched back to back from the calling thread.  The reference's caller shape: one QPSolver per problem
// (/root/reference/include/solvers/qp.hpp:217-247, src/sqp.cpp:210-242) — here N of them across G GPUs.
#pragma once
#include <thread>
#include <vector>

#include "qp.hpp"

namespace qp_solver {
SQP_HIP_INLINE_SUPPORTED namespace supported {

template <typename SCALAR>
class MultiGpuBatchQPSolver {
   public:
    using Scalar = SCALAR;
    using Single = BatchQPSolver<Scalar>;
    using Settings = QPSolverSettings<Scalar>;
    using Info = QPSolverInfo<Scalar>;
    using Batch = typename Single::Batch;

    // num_devices <= 0: every visible device.  `root` = the device the result records are gathered on.
    MultiGpuBatchQPSolver(int n, int m, long long batch, int num_devices = 0, int flags = 0, int root = 0)
        : n_(n), m_(m), batch_(batch), root_(root) {
        int G = num_devices > 0 ? num_devices : sqph_device_count();
        if (G <= 0) throw std::runtime_error("MultiGpuBatchQPSolver: no HIP device visible (this library has no CPU path)");
        std::vector<int> devices;
        for (int g = 0; g < G; g++) devices.push_back(g);
        init(devices, flags);
    }
    // Explicit placement: shard g of the contiguous split lives on devices[g].  A device may be listed more than once — several
    // shards per GPU, each with its own handle and stream (their host-to-device copies and kernels overlap; it is also how the
    // sharding and gather logic is exercised on a one-GPU machine).
    // gather_flags: 0, SQPH_GATHER_RCCL_ALWAYS or SQPH_GATHER_NO_RCCL (sqph_gather_create_ex)
    MultiGpuBatchQPSolver(int n, int m, long long batch, const std::vector<int> &devices, int flags = 0, int root = 0, int gather_flags = 0)
        : n_(n), m_(m), batch_(batch), root_(root), gather_flags_(gather_flags) {
        if (devices.empty()) throw std::runtime_error("MultiGpuBatchQPSolver: empty device list");
        init(devices, flags);
    }
    ~MultiGpuBatchQPSolver() {
        sqph_gather_destroy(gather_);
        for (auto *p : parts_) delete p;
    }
    MultiGpuBatchQPSolver(const MultiGpuBatchQPSolver &) = delete;
    MultiGpuBatchQPSolver &operator=(const MultiGpuBatchQPSolver &) = delete;

    int num_devices() const { return (int)parts_.size(); }
    long long shard_begin(int g) const { return lo_[g]; }
    long long shard_end(int g) const { return hi_[g]; }
    Single &shard(int g) { return *parts_[g]; }
    Settings &settings() { return settings_; }

    // the whole batch in host memory: QP b at base + b * stride (stride 0 = shared by the batch)
    Batch packed(const Scalar *P, const Scalar *q, const Scalar *A, const Scalar *l, const Scalar *u) const {
        return Batch{(int)batch_, SQPH_HOST, P, q, A, l, u, (long long)n_ * n_, n_, (long long)m_ * n_, m_, m_};
    }
    void setup_solve(const Batch &b) { run_host(&Single::setup_solve, b); }
    void setup(const Batch &b) { run_host(&Single::setup, b); }
    void update_qp(const Batch &b) { run_host(&Single::update_qp, b); }
    void solve(const Batch &b) { run_host(&Single::solve, b); }
    // device-resident shards: shards[g] describes device g's block (pointers on device g, batch = its block size)
    void setup_solve_device(const std::vector<Batch> &shards) {
        if ((int)shards.size() != num_devices()) throw std::runtime_error("setup_solve_device: one Batch per device");
        std::vector<sqph_solver *> srcs;
        std::vector<long long> offs;
        std::vector<int> cnts;
        for (int g = 0; g < num_devices(); g++) {
            parts_[g]->settings() = settings_;
            parts_[g]->setup_solve(shards[g]);  // asynchronous: returns once the launch is enqueued on device g's stream
            srcs.push_back(parts_[g]->handle());
            offs.push_back(lo_[g]);
            cnts.push_back((int)(hi_[g] - lo_[g]));
        }
        // every shard's records in ONE grouped post, behind the solves (one RCCL group, the root receiving on its one stream).
        // (an error is recorded on the FAILING shard's handle and in the calling thread's global error: the latter is reported)
        detail::check(sqph_gather_post_many(gather_, srcs.data(), offs.data(), cnts.data(), (int)srcs.size()), nullptr, "sqph_gather_post_many");
        fetched_ = false;
    }

    const Scalar *primal_solution(long long b) { fetch(); return &x_[(size_t)b * n_]; }
    const Scalar *dual_solution(long long b) { fetch(); return &y_[(size_t)b * m_]; }
    const Info &info(long long b) { fetch(); return info_[(size_t)b]; }
    // gathered records on the root device (fp64 x [batch][n], y [batch][m], sqph_info [batch]); waits for the copies
    // "rccl" (shards on other devices than the root's: grouped ncclSend / ncclRecv over xGMI) or "peer-copy"
    const char *gather_transport() const { return sqph_gather_transport(gather_); }
    void gathered_device(void **x, void **y, sqph_info **info) { detail::check(sqph_gather_device_ptrs(gather_, x, y, info), nullptr, "sqph_gather_device_ptrs"); }

   private:
    void init(std::vector<int> devices, int flags) {
        if ((long long)devices.size() > batch_) devices.resize((size_t)batch_);
        const int G = (int)devices.size();
        for (int g = 0; g < G; g++) {
            long long lo, hi;
            sqph_shard_bounds(batch_, G, g, &lo, &hi);
            lo_.push_back(lo);
            hi_.push_back(hi);
            parts_.emplace_back(new Single(n_, m_, (int)(hi - lo), devices[(size_t)g], flags));
            detail::check(sqph_own_stream(parts_.back()->handle()), parts_.back()->handle(), "sqph_own_stream");
        }
        detail::check(sqph_gather_create_ex(&gather_, root_, n_, m_, batch_, gather_flags_), nullptr, "sqph_gather_create");
        x_.resize((size_t)batch_ * n_);
        y_.resize((size_t)batch_ * (m_ > 0 ? m_ : 1));
        raw_.resize((size_t)batch_);
        info_.resize((size_t)batch_);
    }
    void post(int g) {
        detail::check(sqph_gather_post(gather_, parts_[g]->handle(), lo_[g], (int)(hi_[g] - lo_[g])), parts_[g]->handle(), "sqph_gather_post");
    }
    template <typename F>
    void run_host(F fn, const Batch &b) {
        if (b.batch != batch_ || b.memspace != SQPH_HOST) throw std::runtime_error("MultiGpuBatchQPSolver: a host-memspace Batch of the full size is expected");
        std::vector<std::thread> th;
        std::vector<std::string> errs(parts_.size());
        for (int g = 0; g < num_devices(); g++) {
            th.emplace_back([&, g]() {
                try {
                    Batch s = b;
                    s.batch = (int)(hi_[g] - lo_[g]);
                    s.P = b.P + lo_[g] * b.stride_P; s.q = b.q + lo_[g] * b.stride_q; s.A = b.A + lo_[g] * b.stride_A;
                    s.l = b.l + lo_[g] * b.stride_l; s.u = b.u + lo_[g] * b.stride_u;
                    parts_[g]->settings() = settings_;
                    (parts_[g]->*fn)(s);
                    post(g);
                } catch (const std::exception &e) {
                    errs[g] = e.what();
                }
            });
        }
        for (auto &t : th) t.join();
        for (auto &e : errs)
            if (!e.empty()) throw std::runtime_error(e);
        fetched_ = false;
    }
    void fetch() {
        if (fetched_) return;
        detail::check(sqph_gather_fetch(gather_, detail::dtype_of<Scalar>::value, x_.data(), m_ ? y_.data() : nullptr, raw_.data()), nullptr, "sqph_gather_fetch");
        for (size_t b = 0; b < raw_.size(); b++) {
            info_[b].status = (QPSolverStatus)raw_[b].status;
            info_[b].iter = raw_[b].iter;
            info_[b].rho_updates = raw_[b].rho_updates;
            info_[b].rho_estimate = (Scalar)raw_[b].rho_estimate;
            info_[b].res_prim = (Scalar)raw_[b].res_prim;
            info_[b].res_dual = (Scalar)raw_[b].res_dual;
        }
        fetched_ = true;
    }

    int n_, m_;
    long long batch_;
    int root_;
    bool fetched_ = true;
    Settings settings_;
    std::vector<Single *> parts_;
    std::vector<long long> lo_, hi_;
    sqph_gather *gather_ = nullptr;
    int gather_flags_ = 0;
    std::vector<Scalar> x_, y_;
    std::vector<sqph_info> raw_;
    std::vector<Info> info_;
};

}  // namespace supported
}  // namespace qp_solver
