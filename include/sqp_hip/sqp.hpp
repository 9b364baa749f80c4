// sqp.hpp — batched host-side SQP driver on top of the GPU QP-subproblem solver (BASELINE config 4).
//
// The reference's SQP outer loop (src/sqp.cpp:43-101), QP construction + damped BFGS + posdef repair
// (src/sqp.cpp:139-208, include/solvers/bfgs.hpp:14-41), second-order correction (src/sqp.cpp:244-276),
// l1-merit line search (src/sqp.cpp:277-308) and termination (src/sqp.cpp:124-131, 329-343) stay on the
// host, exactly as BASELINE.json's north_star asks; what changes is that N problem instances advance in
// lock-step and all their QP subproblems of an outer iteration go to the GPU in ONE sqph_setup_solve call
// (finished instances are compacted out of the batch).  Per instance the arithmetic is the reference's.
//
//   reference                                         here
//   sqp::NonLinearProblem<Scalar>  sqp.hpp:62-76       sqp::NonLinearProblem<Scalar> (raw pointers, Jc column-major)
//   sqp::sqp_settings_t<Scalar>    sqp.hpp:13-31       sqp::sqp_settings_t<Scalar>
//   sqp::Info / Status             sqp.hpp:33-38       sqp::Info / Status
//   sqp::SQP<Scalar>::solve        sqp.cpp:26-41       sqp::BatchSQP<Scalar>::solve (N instances)
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <exception>
#include <functional>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

#include "qp.hpp"

// The types below live in sqp::raw.  By default that namespace is inline (sqp::NonLinearProblem, sqp::BatchSQP, ...); the drop-in
// header for the reference's SQP class (compat/solvers/sqp.hpp) defines SQP_HIP_SQP_DROPIN, which makes it a plain namespace so
// that sqp::NonLinearProblem / sqp_settings_t / Info / SQP can carry the reference's Eigen-typed definitions.
#ifdef SQP_HIP_SQP_DROPIN
#define SQP_HIP_INLINE_RAW
#else
#define SQP_HIP_INLINE_RAW inline
#endif

namespace sqp {
SQP_HIP_INLINE_RAW namespace raw {

template <typename Scalar_ = double>
struct NonLinearProblem {
    using Scalar = Scalar_;
    int num_var = 0;
    int num_constr = 0;
    virtual void objective(const Scalar *x, Scalar &obj) = 0;
    virtual void objective_linearized(const Scalar *x, Scalar *grad, Scalar &obj) = 0;
    virtual void constraint(const Scalar *x, Scalar *c, Scalar *l, Scalar *u) = 0;
    // Jc: num_constr x num_var, column-major
    virtual void constraint_linearized(const Scalar *x, Scalar *Jc, Scalar *c, Scalar *l, Scalar *u) = 0;
    virtual ~NonLinearProblem() {}
};

template <typename Scalar>
struct sqp_settings_t {
    Scalar tau = 0.5;
    Scalar eta = 0.25;
    Scalar rho = 0.5;
    Scalar eps_prim = 1e-4;
    Scalar eps_dual = 1e-4;
    int max_iter = 100;
    int line_search_max_iter = 20;
    bool second_order_correction = false;
    // NOT in the reference's struct.  Off (default): every QP subproblem is set up from scratch (x, z, y zeroed) exactly as the
    // reference's run_solve_qp does — setup(); solve(), src/sqp.cpp:221-222 — and the trajectories are the reference's.  On: from the
    // second outer iteration on the subproblems go through update_qp(); solve() (src/qp.cpp:46-62), i.e. the ADMM iterates of an
    // instance's previous subproblem are the starting point of its next one — what the reference's `warm_start = true`
    // (src/sqp.cpp:16) asks for and its setup() call undoes.  Fewer ADMM iterations per outer iteration; the iterates differ from
    // the reference's within the subproblems' tolerance (eps 1e-4), the known answers are reached all the same.
    bool warm_start_qp = false;
};

typedef enum { SOLVED, MAX_ITER_EXCEEDED, INVALID_SETTINGS } Status;

struct Info {
    int iter = 0;
    int qp_solver_iter = 0;
    Status status = MAX_ITER_EXCEEDED;
};

// Damped BFGS update ("Procedure 18.2", Nocedal & Wright) of the column-major n x n matrix B with step s and gradient change y:
// the reference's BFGS_update (include/solvers/bfgs.hpp:14-41), statement for statement.  Bs and r are n-vectors of scratch.
template <typename Scalar>
inline void bfgs_update(Scalar *B, int n, const Scalar *s, const Scalar *y, Scalar *Bs, Scalar *r) {
    Scalar sBs = 0, sy = 0, sr;
    for (int i = 0; i < n; i++) {
        Scalar a = 0;
        for (int j = 0; j < n; j++) a += B[(size_t)j * n + i] * s[j];
        Bs[i] = a;
    }
    for (int i = 0; i < n; i++) {
        sBs += s[i] * Bs[i];
        sy += s[i] * y[i];
    }
    if (sy < 0.2 * sBs) {
        const Scalar theta = 0.8 * sBs / (sBs - sy);
        for (int i = 0; i < n; i++) r[i] = theta * y[i] + (1 - theta) * Bs[i];
        sr = theta * sy + (1 - theta) * sBs;
    } else {
        for (int i = 0; i < n; i++) r[i] = y[i];
        sr = sy;
    }
    if (sr < std::numeric_limits<Scalar>::epsilon()) return;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) B[(size_t)j * n + i] += -Bs[i] * Bs[j] / sBs + r[i] * r[j] / sr;
}

// Host threads for the per-instance phases of the batched driver (linearisation + BFGS + QP assembly, second-order correction,
// line search): the instances are independent, every one keeps the reference's arithmetic and order, so the results do not
// depend on the thread count.  Workers spin briefly between the phases of an outer iteration (they are ~20 us apart) and yield /
// sleep when idle for longer.
class HostPool {
   public:
    explicit HostPool(int threads) { resize(threads); }
    ~HostPool() { resize(1); }
    HostPool(const HostPool &) = delete;
    HostPool &operator=(const HostPool &) = delete;
    int threads() const { return (int)workers_.size() + 1; }
    void resize(int threads) {
        if (threads < 1) threads = 1;
        if (threads == this->threads()) return;
        stop_.store(true, std::memory_order_release);
        gen_.fetch_add(1, std::memory_order_release);
        for (auto &t : workers_) t.join();
        workers_.clear();
        stop_.store(false, std::memory_order_release);
        // the generation a worker starts from is read HERE: read inside the new thread, it could already be the first job's
        const unsigned start = gen_.load(std::memory_order_acquire);
        for (int i = 1; i < threads; i++) workers_.emplace_back([this, start] { worker(start); });
    }
    // body(lo, hi) over [0, n) in chunks; the calling thread takes part
    void parallel_for(int n, const std::function<void(int, int)> &body) {
        if (workers_.empty() || n < 2 * CHUNK) {
            if (n > 0) body(0, n);
            return;
        }
        job_ = &body;
        n_ = n;
        next_.store(0, std::memory_order_relaxed);
        pending_.store((int)workers_.size(), std::memory_order_relaxed);
        failed_.store(false, std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_release);
        work();
        // (a worker may be inside its 100 us idle sleep: spin briefly, then give the core away instead of burning it)
        for (unsigned spins = 0; pending_.load(std::memory_order_acquire) != 0; spins++) {
            if (spins < 20000) cpu_relax();
            else std::this_thread::yield();
        }
        // an exception thrown by a callback on any thread (the user's NLP methods run here) surfaces from this call, like in the
        // serial driver — the first one wins, the other chunks of the job are skipped
        if (failed_.load(std::memory_order_acquire)) {
            std::exception_ptr e = error_;
            error_ = nullptr;
            std::rethrow_exception(e);
        }
    }

   private:
    static constexpr int CHUNK = 16;
    static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    void work() {
        for (;;) {
            const int lo = next_.fetch_add(CHUNK, std::memory_order_relaxed);
            if (lo >= n_) break;
            if (failed_.load(std::memory_order_relaxed)) continue;  // drain the remaining chunks without running them
            try {
                (*job_)(lo, lo + CHUNK < n_ ? lo + CHUNK : n_);
            } catch (...) {
                std::lock_guard<std::mutex> lk(err_mu_);
                if (!error_) error_ = std::current_exception();
                failed_.store(true, std::memory_order_release);
            }
        }
    }
    void worker(unsigned last) {
        for (;;) {
            unsigned g, spins = 0;
            while ((g = gen_.load(std::memory_order_acquire)) == last) {
                if (++spins < 20000) cpu_relax();
                else if (spins < 40000) std::this_thread::yield();
                else std::this_thread::sleep_for(std::chrono::microseconds(100));
            }
            last = g;
            if (stop_.load(std::memory_order_acquire)) return;
            work();
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> workers_;
    std::atomic<unsigned> gen_{0};
    std::atomic<int> next_{0}, pending_{0};
    std::atomic<bool> stop_{false}, failed_{false};
    std::mutex err_mu_;
    std::exception_ptr error_;
    const std::function<void(int, int)> *job_ = nullptr;
    int n_ = 0;
};

// QPBackend: the batched QP-subproblem solver (setup_solve / info / primal_solution / dual_solution over a packed
// batch).  The product instantiates the default — qp_solver::BatchQPSolver, i.e. libsqp_hip; tests/cpp substitutes a
// backend that calls the CPU oracle to separate "the host driver is exact" from "a QP rounding flipped a line search".
template <typename Scalar_, typename QPBackend = qp_solver::BatchQPSolver<Scalar_>>
class BatchSQP {
   public:
    using Scalar = Scalar_;
    using Problem = NonLinearProblem<Scalar>;
    using Settings = sqp_settings_t<Scalar>;
    static constexpr Scalar DIV_BY_ZERO_REGUL = std::numeric_limits<Scalar>::epsilon();

    // The QP backend keeps its factors resident (SQPH_FLAG_KEEP_FACTOR): the second-order correction re-solves with new bounds
    // only and reuses them (the reference's TODO at src/sqp.cpp:273).
    BatchSQP(int num_var, int num_constr, int batch, int device = 0)
        : n_(num_var), m_(num_constr), batch_(batch), qp_(num_var, num_constr, batch, device, SQPH_FLAG_KEEP_FACTOR), inst_(batch) {
        // QP settings of the reference's SQP constructor, src/sqp.cpp:15-23
        auto &s = qp_.settings();
        s.warm_start = true;
        s.check_termination = 10;
        s.eps_abs = 1e-4;
        s.eps_rel = 1e-4;
        s.max_iter = 100;
        s.adaptive_rho = true;
        s.adaptive_rho_interval = 50;
        s.alpha = 1.6;
        const size_t n = n_, m = m_ > 0 ? m_ : 1, B = batch;
        P_.resize(B * n * n); q_.resize(B * n); A_.resize(B * m * n); l_.resize(B * m); u_.resize(B * m);
        for (auto &I : inst_) I.init(n_, m_);
    }

    Settings &settings() { return settings_; }
    // Host threads for the per-instance phases (default 1).  With more than one, NonLinearProblem objects shared by several
    // instances are called concurrently (for different instances) and must be stateless; the results do not depend on the count.
    // The phases run on the calling thread while a trace or step callback is installed.
    void set_host_threads(int threads) { pool_.resize(threads); }
    int host_threads() const { return pool_.threads(); }
    qp_solver::QPSolverSettings<Scalar> &qp_settings() { return qp_.settings(); }
    QPBackend &qp_backend() { return qp_; }

    // probs[i] is the NLP of instance i (several entries may point to one stateless object).
    // X0: [batch][num_var], Lambda0: [batch][num_constr] (nullptr = zeros, as SQP::solve(prob) does).
    void solve(const std::vector<Problem *> &probs, const Scalar *X0, const Scalar *Lambda0) {
        const int n = n_, m = m_;
        std::vector<int> live;
        for (int i = 0; i < batch_; i++) {
            Inst &I = inst_[i];
            for (int k = 0; k < n; k++) I.x[k] = X0 ? X0[(size_t)i * n + k] : Scalar(0);
            for (int k = 0; k < m; k++) I.lambda[k] = Lambda0 ? Lambda0[(size_t)i * m + k] : Scalar(0);
            I.info = Info();
            live.push_back(i);
        }
        // warm-started subproblems keep their state in the QP backend by SLOT: instance i stays in slot i for the whole solve
        // (finished instances are not compacted out; their slot is re-solved from its own solution, which ends at the first check)
        const bool warm = settings_.warm_start_qp;
        int iter;
        for (iter = 1; iter <= settings_.max_iter && !live.empty(); iter++) {
            // ---- solve_qp (src/sqp.cpp:139-199) for every live instance: build the QP on the host ...
            phase((int)live.size(), [&](int k_lo, int k_hi) {
            for (size_t k = (size_t)k_lo; k < (size_t)k_hi; k++) {
                Inst &I = inst_[live[k]];
                Problem &prob = *probs[live[k]];
                I.info.iter = iter;
                prob.objective_linearized(I.x.data(), I.grad_obj.data(), I.obj);
                prob.constraint_linearized(I.x.data(), I.Jac.data(), I.constr.data(), I.l.data(), I.u.data());
                for (int a = 0; a < n; a++) I.delta_grad_L[a] = -I.grad_L[a];
                for (int j = 0; j < n; j++) {
                    Scalar acc = 0;
                    for (int a = 0; a < m; a++) acc += I.Jac[(size_t)j * m + a] * I.lambda[a];
                    I.grad_L[j] = I.grad_obj[j] + acc;
                }
                if (iter == 1) {
                    for (int j = 0; j < n; j++)
                        for (int a = 0; a < n; a++) I.Hess[(size_t)j * n + a] = (a == j) ? Scalar(1) : Scalar(0);
                } else {
                    for (int a = 0; a < n; a++) I.delta_grad_L[a] += I.grad_L[a];
                    bfgs_update(I);
                }
                if (!is_posdef(I)) {  // src/sqp.cpp:172-181
                    Scalar tau = 1e-3;
                    while (!is_posdef(I)) {
                        for (int a = 0; a < n; a++) I.Hess[(size_t)a * n + a] += tau;
                        tau *= 10;
                    }
                }
                for (int a = 0; a < m; a++) {
                    I.ql[a] = I.l[a] - I.constr[a];
                    I.qu[a] = I.u[a] - I.constr[a];
                }
                pack(warm ? (size_t)live[k] : k, I);
            }
            });
            // ---- ... and run_solve_qp (src/sqp.cpp:210-242) for all of them in one launch
            run_qp(live, false, warm, iter == 1);
            if (settings_.second_order_correction) {  // src/sqp.cpp:244-276
                phase((int)live.size(), [&](int k_lo, int k_hi) {
                for (size_t k = (size_t)k_lo; k < (size_t)k_hi; k++) {
                    Inst &I = inst_[live[k]];
                    Problem &prob = *probs[live[k]];
                    for (int a = 0; a < n; a++) I.x_step[a] = I.x[a] + I.p[a];
                    prob.constraint(I.x_step.data(), I.c_step.data(), I.l.data(), I.u.data());
                    for (int a = 0; a < m; a++) {
                        Scalar acc = 0;
                        for (int j = 0; j < n; j++) acc += I.Jac[(size_t)j * m + a] * I.p[j];
                        const Scalar d = I.c_step[a] - acc;
                        I.ql[a] = I.l[a] - d;
                        I.qu[a] = I.u[a] - d;
                    }
                    pack(warm ? (size_t)live[k] : k, I);
                }
                });
                run_qp(live, /*same_matrices=*/true, warm, false);
            }
            // ---- step, line search, termination (src/sqp.cpp:76-96)
            std::vector<int> still;
            done_.assign(live.size(), 0);
            phase((int)live.size(), [&](int k_lo, int k_hi) {
            for (size_t k = (size_t)k_lo; k < (size_t)k_hi; k++) {
                Inst &I = inst_[live[k]];
                Problem &prob = *probs[live[k]];
                for (int a = 0; a < m; a++) I.p_lambda[a] -= I.lambda[a];
                const Scalar alpha = line_search(I, prob);
                if (trace_) {
                    const Scalar *const qpv[6] = {I.Hess.data(), I.grad_obj.data(), I.Jac.data(), I.ql.data(), I.qu.data(), I.lambda.data()};
                    trace_(trace_user_, live[k], iter, I.p.data(), I.p_lambda.data(), alpha, I.info.qp_solver_iter, qpv);
                }
                for (int a = 0; a < n; a++) I.x[a] += alpha * I.p[a];
                for (int a = 0; a < m; a++) I.lambda[a] += alpha * I.p_lambda[a];
                for (int a = 0; a < n; a++) I.step_prev[a] = alpha * I.p[a];
                if (step_) step_(step_user_, live[k], iter, I.x.data(), I.lambda.data());
                const Scalar primal_step_norm = alpha * inf_norm(I.p), dual_step_norm = alpha * inf_norm(I.p_lambda);
                if (primal_step_norm <= settings_.eps_prim && dual_step_norm <= settings_.eps_dual &&
                    max_constraint_violation(I, prob) <= settings_.eps_prim) {
                    I.info.status = SOLVED;
                    done_[k] = 1;
                }
            }
            });
            for (size_t k = 0; k < live.size(); k++)
                if (!done_[k]) still.push_back(live[k]);
            live.swap(still);
        }
        for (int i : live) {  // exhausted: src/sqp.cpp:98-100 (iter == max_iter + 1)
            inst_[i].info.status = MAX_ITER_EXCEEDED;
            inst_[i].info.iter = settings_.max_iter + 1;
        }
    }

    // Called for every live instance right after its step has been taken (x, lambda updated) — where the reference calls
    // settings.iteration_callback (src/sqp.cpp:88-90).
    typedef void (*step_fn)(void *user, int instance, int iter, const Scalar *x, const Scalar *lambda);
    void set_step_callback(step_fn f, void *user) { step_ = f; step_user_ = user; }

    // Per-instance trajectory record, called once per outer iteration after the line search (the reference has no such
    // hook; used by the parity tests to locate the first outer iteration at which two runs separate).
    // `qp` = the five arrays (P, q, A, l, u; column-major) of the last QP subproblem of that outer iteration, then the
    // multiplier estimate lambda the dual step was taken from (QP dual = p_lambda + lambda).
    typedef void (*trace_fn)(void *user, int instance, int iter, const Scalar *p, const Scalar *p_lambda, Scalar alpha, int qp_iter,
                             const Scalar *const qp[6]);
    void set_trace(trace_fn f, void *user) { trace_ = f; trace_user_ = user; }

    const Scalar *primal_solution(int i) const { return inst_[i].x.data(); }
    const Scalar *dual_solution(int i) const { return inst_[i].lambda.data(); }
    const Info &info(int i) const { return inst_[i].info; }
    int qp_launches() const { return launches_; }
    double qp_backend_ms() const { return qp_ms_; }  // wall time spent inside the QP backend (staging, launch, fetch)

   private:
    struct Inst {
        std::vector<Scalar> x, lambda, step_prev, grad_L, delta_grad_L, Hess, grad_obj, Jac, constr, l, u;
        std::vector<Scalar> p, p_lambda, ql, qu, x_step, c_step, work, Bs, r;
        Scalar obj = 0;
        Info info;
        void init(int n, int m) {
            const size_t mm = m > 0 ? m : 1;
            x.assign(n, 0); lambda.assign(mm, 0); step_prev.assign(n, 0); grad_L.assign(n, 0); delta_grad_L.assign(n, 0);
            Hess.assign((size_t)n * n, 0); grad_obj.assign(n, 0); Jac.assign(mm * n, 0); constr.assign(mm, 0); l.assign(mm, 0); u.assign(mm, 0);
            p.assign(n, 0); p_lambda.assign(mm, 0); ql.assign(mm, 0); qu.assign(mm, 0); x_step.assign(n, 0); c_step.assign(mm, 0);
            work.assign((size_t)n * n, 0); Bs.assign(n, 0); r.assign(n, 0);
        }
    };

    // a per-instance phase: on the pool unless a callback is installed (callbacks are invoked in instance order, from one thread)
    template <typename F>
    void phase(int count, F &&body) {
        if (trace_ || step_ || pool_.threads() == 1) body(0, count);
        else pool_.parallel_for(count, body);
    }
    void pack(size_t k, const Inst &I) {
        const size_t n = n_, m = m_;
        std::copy(I.Hess.begin(), I.Hess.end(), P_.begin() + k * n * n);
        std::copy(I.grad_obj.begin(), I.grad_obj.end(), q_.begin() + k * n);
        std::copy(I.Jac.begin(), I.Jac.begin() + m * n, A_.begin() + k * m * n);
        std::copy(I.ql.begin(), I.ql.begin() + m, l_.begin() + k * m);
        std::copy(I.qu.begin(), I.qu.begin() + m, u_.begin() + k * m);
    }
    void run_qp(const std::vector<int> &live, bool same_matrices = false, bool warm = false, bool first = true) {
        const auto t0 = std::chrono::steady_clock::now();
        // warm: every slot of the batch (slot = instance), cold set-up only in the first outer iteration
        const auto batch = qp_.packed(warm ? batch_ : (int)live.size(), P_.data(), q_.data(), A_.data(), l_.data(), u_.data());
        if (warm && !first) qp_.update_solve(batch);
        else if (same_matrices && !warm) qp_.setup_solve_reuse(batch);  // P, A of the call before (same live set, same order): factor reuse
        else qp_.setup_solve(batch);
        (void)qp_.info(0);  // fetch the results (one packed D2H)
        qp_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        launches_++;
        for (size_t k = 0; k < live.size(); k++) {
            Inst &I = inst_[live[k]];
            const int slot = warm ? live[k] : (int)k;
            const auto &qi = qp_.info(slot);
            I.info.qp_solver_iter += qi.iter;
            if (qi.status == qp_solver::NUMERICAL_ISSUES) continue;  // prim/dual left untouched, src/sqp.cpp:226-229
            for (int a = 0; a < n_; a++) I.p[a] = qp_.primal_solution(slot)[a];
            for (int a = 0; a < m_; a++) I.p_lambda[a] = qp_.dual_solution(slot)[a];
        }
    }
    // Eigen::LLT-style test, src/sqp.cpp:115-122
    bool is_posdef(Inst &I) const {
        const int n = n_;
        std::vector<Scalar> &w = I.work;
        w = I.Hess;
        for (int k = 0; k < n; k++) {
            Scalar x = w[(size_t)k * n + k];
            for (int j = 0; j < k; j++) x -= w[(size_t)j * n + k] * w[(size_t)j * n + k];
            if (!(x > Scalar(0))) return false;
            x = std::sqrt(x);
            w[(size_t)k * n + k] = x;
            for (int i = k + 1; i < n; i++) {
                Scalar v = w[(size_t)k * n + i];
                for (int j = 0; j < k; j++) v -= w[(size_t)j * n + i] * w[(size_t)j * n + k];
                w[(size_t)k * n + i] = v / x;
            }
        }
        return true;
    }
    void bfgs_update(Inst &I) const { raw::bfgs_update(I.Hess.data(), n_, I.step_prev.data(), I.delta_grad_L.data(), I.Bs.data(), I.r.data()); }
    Scalar constraint_norm(const Inst &I) const {  // src/sqp.cpp:310-318
        Scalar c_l1 = DIV_BY_ZERO_REGUL, a = 0, b = 0;
        for (int i = 0; i < m_; i++) a += (I.l[i] - I.constr[i]) > Scalar(0) ? (I.l[i] - I.constr[i]) : Scalar(0);
        for (int i = 0; i < m_; i++) b += (I.constr[i] - I.u[i]) > Scalar(0) ? (I.constr[i] - I.u[i]) : Scalar(0);
        c_l1 += a;
        c_l1 += b;
        return c_l1;
    }
    Scalar line_search(Inst &I, Problem &prob) const {  // src/sqp.cpp:277-308
        const int n = n_;
        const Scalar constr_l1 = constraint_norm(I);
        Scalar gp = 0, pHp = 0;
        for (int i = 0; i < n; i++) gp += I.grad_obj[i] * I.p[i];
        for (int i = 0; i < n; i++) {
            Scalar a = 0;
            for (int j = 0; j < n; j++) a += I.Hess[(size_t)j * n + i] * I.p[j];
            pHp += I.p[i] * a;
        }
        const Scalar mu = (gp + 0.5 * pHp) / ((1 - settings_.rho) * constr_l1);
        const Scalar phi_l1 = I.obj + mu * constr_l1;
        const Scalar Dp_phi_l1 = gp - mu * constr_l1;
        Scalar alpha = 1.0;
        for (int i = 1; i < settings_.line_search_max_iter; i++) {
            Scalar obj_step;
            for (int k = 0; k < n; k++) I.x_step[k] = I.x[k] + alpha * I.p[k];
            prob.objective(I.x_step.data(), obj_step);
            prob.constraint(I.x_step.data(), I.constr.data(), I.l.data(), I.u.data());
            const Scalar phi_l1_step = obj_step + mu * constraint_norm(I);
            if (phi_l1_step <= phi_l1 + alpha * settings_.eta * Dp_phi_l1) break;
            alpha = settings_.tau * alpha;
        }
        return alpha;
    }
    Scalar inf_norm(const std::vector<Scalar> &v) const {
        Scalar r = 0;
        for (size_t i = 0; i < v.size(); i++) r = std::fabs(v[i]) > r ? std::fabs(v[i]) : r;
        return r;
    }
    Scalar max_constraint_violation(Inst &I, Problem &prob) const {  // src/sqp.cpp:329-343
        Scalar c_max = 0;
        prob.constraint(I.x.data(), I.constr.data(), I.l.data(), I.u.data());
        if (m_ > 0) {
            Scalar a = -std::numeric_limits<Scalar>::infinity(), b = a;
            for (int i = 0; i < m_; i++) a = (I.l[i] - I.constr[i]) > a ? (I.l[i] - I.constr[i]) : a;
            for (int i = 0; i < m_; i++) b = (I.constr[i] - I.u[i]) > b ? (I.constr[i] - I.u[i]) : b;
            c_max = std::fmax(c_max, a);
            c_max = std::fmax(c_max, b);
        }
        return c_max;
    }

    int n_, m_, batch_;
    int launches_ = 0;
    double qp_ms_ = 0;
    Settings settings_;
    QPBackend qp_;
    trace_fn trace_ = nullptr;
    step_fn step_ = nullptr;
    void *step_user_ = nullptr;
    void *trace_user_ = nullptr;
    std::vector<Inst> inst_;
    std::vector<Scalar> P_, q_, A_, l_, u_;
    std::vector<char> done_;
    HostPool pool_{1};
};

}  // namespace raw
}  // namespace sqp
