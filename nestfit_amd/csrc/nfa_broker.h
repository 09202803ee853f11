// nfa_broker.h -- callback-coalescing broker, host code only.
//
// MultiNest hands LogLike one point at a time (nestfit/core/cmultinest.pxd:27-28 through
// mn_loglikelihood, nestfit/core/core.pyx:622-624); the GPU wants batches.  The broker lets
// many sampler threads call a blocking, LogLike-shaped entry point; concurrent calls are
// gathered into one nfa_runner_loglike_batch launch.  No service thread: the first caller
// of a generation is its leader, waits until `n_clients` requests (or `max_batch`) are
// queued or `max_wait_us` elapse, runs the batch and wakes the others.  Results are bitwise
// those of a direct call (the engine's per-item results do not depend on the batch).
#pragma once
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

struct nfa_broker {
    nfa_runner *r = nullptr;
    int      max_batch = 4096;
    int      n_clients = 0;            // 0 = unknown: the leader always waits max_wait_us
    int64_t  max_wait_us = 200;
    // one generation = the requests that will share a launch; followers sleep on their own
    // generation's condition variable, so a finished batch wakes only its members
    struct Gen { std::mutex m; std::condition_variable cv; bool done = false; };
    struct Req { double *cube; int pix; double lnl; int rc; };
    std::mutex m;                      // queue state
    std::condition_variable cv_full;   // leader: enough requests queued
    std::vector<Req *> queue;
    std::shared_ptr<Gen> gen;          // generation being filled
    bool leader_present = false;
    std::mutex run_m;                  // the runner is used by one batch at a time
    std::vector<double> U, lnL;        // staging (guarded by run_m)
    std::vector<int32_t> pix;
    uint64_t n_batches = 0, n_evals = 0, max_seen = 0;
};

// nfa_broker_client (`context` of nfa_broker_callback) is defined in nestfit_amd.h

extern "C" {

int nfa_broker_create(nfa_broker **out, nfa_runner *r, int max_batch, int64_t max_wait_us, int n_clients) {
    if (!out || !r) return fail(NFA_ERR_ARG, "null argument");
    if (max_batch < 1 || max_wait_us < 0 || n_clients < 0) return fail(NFA_ERR_ARG, "bad broker limits");
    nfa_broker *b = new nfa_broker();
    b->r = r; b->max_batch = max_batch; b->max_wait_us = max_wait_us; b->n_clients = n_clients;
    *out = b;
    return NFA_OK;
}

int nfa_broker_destroy(nfa_broker *b) {
    if (!b) return NFA_OK;
    {
        std::unique_lock<std::mutex> lk(b->m);
        if (!b->queue.empty() || b->leader_present) return fail(NFA_ERR_STATE, "broker still has callers");
    }
    delete b;
    return NFA_OK;
}

int nfa_broker_set_clients(nfa_broker *b, int n_clients) {
    if (!b || n_clients < 0) return fail(NFA_ERR_ARG, "bad argument");
    std::unique_lock<std::mutex> lk(b->m);
    b->n_clients = n_clients;
    b->cv_full.notify_all();           // a waiting leader re-evaluates its target
    return NFA_OK;
}

static bool broker_ready(const nfa_broker *b) {
    const size_t target = b->n_clients > 0 ? (size_t)std::min(b->n_clients, b->max_batch) : (size_t)b->max_batch;
    return b->queue.size() >= target;
}

// Blocking; any thread.  `cube` (ndim doubles, unit cube) is overwritten with the physical
// parameters like AmmoniaRunner.c_loglikelihood (ammonia.pyx:423-432); pix < 0 = the runner's
// single pixel.
int nfa_broker_loglike(nfa_broker *b, int32_t pix, double *cube, double *lnew) {
    if (!b || !cube || !lnew) return fail(NFA_ERR_ARG, "null argument");
    nfa_broker::Req rq{cube, pix, NAN, NFA_OK};
    std::unique_lock<std::mutex> lk(b->m);
    b->queue.push_back(&rq);
    if (b->leader_present) {
        std::shared_ptr<nfa_broker::Gen> g = b->gen;
        if (broker_ready(b)) b->cv_full.notify_one();
        lk.unlock();
        std::unique_lock<std::mutex> gl(g->m);
        g->cv.wait(gl, [&] { return g->done; });
        *lnew = rq.lnl;
        return rq.rc == NFA_OK ? NFA_OK : fail(rq.rc, "broker batch failed");
    }
    b->leader_present = true;
    std::shared_ptr<nfa_broker::Gen> g = std::make_shared<nfa_broker::Gen>();
    b->gen = g;
    if (b->max_wait_us > 0 && !broker_ready(b))
        b->cv_full.wait_for(lk, std::chrono::microseconds(b->max_wait_us), [&] { return broker_ready(b); });
    std::vector<nfa_broker::Req *> batch;
    batch.swap(b->queue);
    b->leader_present = false;         // the next arrival leads the next generation
    b->gen.reset();
    lk.unlock();

    const int ndim = b->r->ndim;
    const int64_t B = (int64_t)batch.size();
    int rc;
    {
        std::lock_guard<std::mutex> run(b->run_m);
        b->U.resize((size_t)B * ndim); b->lnL.resize((size_t)B); b->pix.resize((size_t)B);
        bool any_pix = false;
        for (int64_t k = 0; k < B; ++k) {
            memcpy(b->U.data() + k * ndim, batch[k]->cube, sizeof(double) * ndim);
            b->pix[k] = batch[k]->pix < 0 ? 0 : batch[k]->pix;
            any_pix |= batch[k]->pix >= 0;
        }
        rc = nfa_runner_loglike_batch(b->r, any_pix ? b->pix.data() : nullptr, b->U.data(), b->lnL.data(), B);
        for (int64_t k = 0; k < B; ++k) {
            if (rc == NFA_OK) memcpy(batch[k]->cube, b->U.data() + k * ndim, sizeof(double) * ndim);
            batch[k]->lnl = rc == NFA_OK ? b->lnL[k] : NAN;
            batch[k]->rc = rc;
        }
        b->n_batches += 1; b->n_evals += (uint64_t)B; b->max_seen = std::max<uint64_t>(b->max_seen, (uint64_t)B);
    }
    {
        std::lock_guard<std::mutex> gl(g->m);
        g->done = true;
    }
    g->cv.notify_all();
    *lnew = rq.lnl;
    return rc;
}

// MultiNest `LogLike` signature; context = nfa_broker_client*.  No error channel: NaN.
void nfa_broker_callback(double *Cube, int *ndim, int *npars, double *lnew, void *ctx) {
    (void)npars;
    nfa_broker_client *c = (nfa_broker_client *)ctx;
    if (!c || !c->broker || !Cube || !lnew || !ndim || *ndim != c->broker->r->ndim) {
        if (lnew) *lnew = NAN;
        return;
    }
    if (nfa_broker_loglike(c->broker, c->pix, Cube, lnew) != NFA_OK) *lnew = NAN;
}

// out[0] batches launched, out[1] evaluations served, out[2] largest batch
int nfa_broker_stats(nfa_broker *b, int64_t *out) {
    if (!b || !out) return fail(NFA_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> run(b->run_m);
    out[0] = (int64_t)b->n_batches; out[1] = (int64_t)b->n_evals; out[2] = (int64_t)b->max_seen;
    return NFA_OK;
}

}  // extern "C"
