// ctm_move: ONE C entry for a whole directional move of the generic CTMRG (reference seam ctm/generic/ctmrg.py:233-283, ctm_MOVE_c:
// "raw tuple sites + C's + T's -> raw tuple nC1[sites] + nC2[sites] + nT[sites]").
//
// Phase A: the projector units of all sites (fused corners -> implicit R^T Rt -> leading-chi triplets -> P, Pt), independent of each
// other (they read the old environment, ctmrg.py:238-275); phase B: the absorb + normalise units, which need the projectors of the
// site AND of its neighbour along the move.  The units of a phase run concurrently on the worker contexts the caller hands in (own
// arena and HIP stream each), from threads of this library: no host-language code runs between the two phases.  Every worker stream
// waits for the work queued on the calling context's stream, and the call returns after every worker stream has drained, so the caller
// sees an ordinary synchronous entry point.
#include "ctm_common.h"
#include <thread>
#include <atomic>
#include <algorithm>
#include <mutex>

namespace {

// shape of the new tensors of one absorb with Y projector columns on an environment of dimension X (see ctm_absorb_x): axis of the
// new bond in nC1 / nC2, and the position of the D^2 leg in nT
struct OutLayout { int c1_new, c2_new, t_d2_axis; };
const OutLayout kOut[4] = {{0, 1, 1}, {0, 0, 2}, {1, 1, 0}, {0, 1, 1}};       // UP, LEFT, DOWN, RIGHT (host layer: _NEW_AX)
const int kOutLeg[4] = {3, 4, 1, 2};

int run_projectors(ctm_ctx* w, int dir, ctm_move_unit* u, int chi, const ctm_trunc_cfg* cfg) {
    return ctm_projectors_4x4_cc(w, dir, u->proj, chi, u->proj_adims, cfg, u->P, u->Pt, u->S, u->basis,
                                 u->use_corner_cache ? u->corner_buf : nullptr, u->use_corner_cache ? u->corner_valid : nullptr);
}

// absorb of one site with the non-zero prefix (y columns) of the projectors: compacted projectors, reduced outputs, zero padding --
// the numbers the full absorb produces (the dropped columns are exact zeros), tests/test_gpu_generic.py
int run_absorb(ctm_ctx* w, int dir, const ctm_move_unit* u, const ctm_move_unit* nb, int chi, int y, int normalize) {
    const double* t10[10] = {u->absorb[0], u->absorb[1], u->absorb[2], u->absorb[3], u->absorb[4], u->absorb[5], u->P, u->Pt, nb->P, nb->Pt};
    const int pcols = (int)std::min<long long>(chi, u->n_rows);          // columns of P, Pt as ctm_projectors_4x4 wrote them (chi unless chi > n)
    if ((int)std::min<long long>(chi, nb->n_rows) != pcols) { w->set_error("ctm_move: a site and its neighbour have projectors of different width (chi > n on one of them)"); return CTM_ERR_UNSUPPORTED; }
    if (y >= pcols) return ctm_absorb_x(w, dir, t10, chi, pcols, u->absorb_adims, normalize, u->nC1, u->nC2, u->nT);
    const int cz = w->cplx ? 2 : 1;
    const long long D2 = (long long)u->absorb_adims[kOutLeg[dir]] * u->absorb_adims[kOutLeg[dir]];
    ArenaScope scope(w);
    double *pc[4], *c1, *c2, *nt;
    const double* full[4] = {u->P, u->Pt, nb->P, nb->Pt};
    const long long nrow[4] = {u->n_rows, u->n_rows, nb->n_rows, nb->n_rows};
    for (int i = 0; i < 4; ++i) {
        CTM_TRY(arena_alloc(w, sizeof(double) * cz * (size_t)nrow[i] * y, (void**)&pc[i]));
        CTM_TRY(copy2d(w, full[i], (long long)cz * chi, pc[i], (long long)cz * y, (int)nrow[i], cz * y));
        t10[6 + i] = pc[i];
    }
    CTM_TRY(arena_alloc(w, sizeof(double) * cz * (size_t)chi * y, (void**)&c1));
    CTM_TRY(arena_alloc(w, sizeof(double) * cz * (size_t)chi * y, (void**)&c2));
    CTM_TRY(arena_alloc(w, sizeof(double) * cz * (size_t)y * y * D2, (void**)&nt));
    CTM_TRY(ctm_absorb_x(w, dir, t10, chi, y, u->absorb_adims, normalize, c1, c2, nt));
    const OutLayout& L = kOut[dir];
    CTM_TRY(fill_f64(w, u->nC1, (size_t)cz * chi * chi, 0.0));
    CTM_TRY(fill_f64(w, u->nC2, (size_t)cz * chi * chi, 0.0));
    CTM_TRY(fill_f64(w, u->nT, (size_t)cz * chi * chi * D2, 0.0));
    auto pad_c = [&](const double* src, double* dst, int new_axis) -> int {
        if (new_axis == 0) return copy2d(w, src, (long long)cz * chi, dst, (long long)cz * chi, y, cz * chi);      // (y, chi) -> first y rows
        return copy2d(w, src, (long long)cz * y, dst, (long long)cz * chi, chi, cz * y);                           // (chi, y) -> first y columns
    };
    CTM_TRY(pad_c(c1, u->nC1, L.c1_new));
    CTM_TRY(pad_c(c2, u->nC2, L.c2_new));
    if (L.t_d2_axis == 1) {          // (y, D2, y) -> (chi, D2, chi)
        for (int i = 0; i < y; ++i)
            CTM_TRY(copy2d(w, nt + (size_t)cz * i * D2 * y, (long long)cz * y, u->nT + (size_t)cz * i * D2 * chi, (long long)cz * chi, (int)D2, cz * y));
    } else if (L.t_d2_axis == 2) {   // (y, y, D2) -> (chi, chi, D2)
        for (int i = 0; i < y; ++i)
            CTM_TRY(copy2d(w, nt + (size_t)cz * i * y * D2, (long long)cz * D2, u->nT + (size_t)cz * i * chi * D2, (long long)cz * D2, y, cz * (int)D2));
    } else {                         // (D2, y, y) -> (D2, chi, chi)
        for (int i = 0; i < (int)D2; ++i)
            CTM_TRY(copy2d(w, nt + (size_t)cz * i * y * y, (long long)cz * y, u->nT + (size_t)cz * i * chi * chi, (long long)cz * chi, y, cz * y));
    }
    CTM_HIP_CHECK(w, hipStreamSynchronize(w->stream));
    return CTM_OK;
}

// units 0 .. n-1 through fn(worker context, unit index), at most nw at a time, one thread per worker context
template <class F>
int for_units(ctm_ctx* ctx, ctm_ctx* const* workers, int nw, int n, F&& fn, std::string* err) {
    if (nw <= 0) {                   // serial, on the calling context itself
        for (int i = 0; i < n; ++i) { const int st = fn(ctx, i); if (st != CTM_OK) return st; }
        return CTM_OK;
    }
    std::atomic<int> next{0}, status{CTM_OK};
    std::vector<std::thread> th;
    std::mutex emu;
    const int nt = std::min(nw, n);
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            ctm_ctx* w = workers[t];
            (void)hipSetDevice(ctx->device);
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= n || status.load() != CTM_OK) break;
                int st;
                try { st = fn(w, i); }
                catch (const std::bad_alloc&) { st = CTM_ERR_NOMEM; w->set_error("ctm_move: host allocation failed in a unit"); }
                catch (const std::exception& e) { st = CTM_ERR_HIP; w->set_error(std::string("ctm_move: C++ exception in a unit: ") + e.what()); }
                catch (...) { st = CTM_ERR_HIP; w->set_error("ctm_move: unknown exception in a unit"); }      // nothing may reach std::terminate from a thread
                if (st != CTM_OK) {
                    int expect = CTM_OK;
                    if (status.compare_exchange_strong(expect, st)) { std::lock_guard<std::mutex> l(emu); *err = w->last_error; }
                }
            }
        });
    for (auto& t : th) t.join();
    return status.load();
}

}  // namespace

extern "C" int ctm_move(ctm_ctx* ctx, ctm_ctx* const* workers, int nworkers, int dir, int nunits, ctm_move_unit* units, int chi,
                        const ctm_trunc_cfg* cfg, int normalize, int skip_zero_columns) {
    return ctm_entry(ctx, "ctm_move", [&]() -> int {
        if (dir < 0 || dir > 3 || nunits < 1 || !units || chi < 1 || nworkers < 0 || (nworkers > 0 && !workers)) { ctx->set_error("ctm_move: bad arguments"); return CTM_ERR_BADARG; }
        for (int i = 0; i < nunits; ++i) {
            const ctm_move_unit& u = units[i];
            if (u.nb < 0 || u.nb >= nunits || !u.P || !u.Pt || !u.S || !u.nC1 || !u.nC2 || !u.nT || u.n_rows < 1) { ctx->set_error("ctm_move: bad unit " + std::to_string(i)); return CTM_ERR_BADARG; }
        }
        for (int t = 0; t < nworkers; ++t) {
            if (!workers[t] || workers[t] == ctx || workers[t]->cplx != ctx->cplx) { ctx->set_error("ctm_move: worker contexts must be distinct contexts of the same dtype"); return CTM_ERR_BADARG; }
            for (int s = 0; s < t; ++s) if (workers[s] == workers[t]) { ctx->set_error("ctm_move: a worker context appears twice"); return CTM_ERR_BADARG; }
        }
        // the workers' streams start after everything queued on the caller's stream (the tensors of the old environment)
        if (nworkers > 0) {
            hipEvent_t ev;
            CTM_HIP_CHECK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            hipError_t e = hipEventRecord(ev, ctx->stream);
            for (int t = 0; t < nworkers && e == hipSuccess; ++t) e = hipStreamWaitEvent(workers[t]->stream, ev, 0);
            (void)hipEventDestroy(ev);
            CTM_HIP_CHECK(ctx, e);
        }
        std::string err;
        // a unit that fails returns mid-pipeline: kernels writing its outputs or its arena may still be queued on its worker stream, and
        // the caller gives the output tensors back to its allocator as soon as it sees the status -- every stream is drained first
        auto drain = [&]() {
            for (int t = 0; t < nworkers; ++t) (void)hipStreamSynchronize(workers[t]->stream);
            (void)hipStreamSynchronize(ctx->stream);
        };
        // phase A: projectors of every site from the old environment
        int st = for_units(ctx, workers, nworkers, nunits, [&](ctm_ctx* w, int i) { return run_projectors(w, dir, &units[i], chi, cfg); }, &err);
        if (st != CTM_OK) { drain(); if (!err.empty()) ctx->set_error("ctm_move, projector unit: " + err); return st; }
        // non-zero projector columns per site (S descending: a prefix); ctm_projectors_4x4 has drained its stream
        std::vector<double> hs(chi);
        for (int i = 0; i < nunits; ++i) {
            const int kc = (int)std::min<long long>(chi, units[i].n_rows);
            CTM_HIP_CHECK(ctx, hipMemcpy(hs.data(), units[i].S, sizeof(double) * kc, hipMemcpyDeviceToHost));
            int nz = 0;
            const double reltol = cfg ? cfg->svd_reltol : 1e-8;
            for (int j = 0; j < kc; ++j) nz += (hs[0] > 0.0 && hs[j] > reltol * hs[0]) ? 1 : 0;
            units[i].ncol = nz;
        }
        // phase B: absorb + normalise every site with its own and its neighbour's projectors
        st = for_units(ctx, workers, nworkers, nunits, [&](ctm_ctx* w, int i) {
            const ctm_move_unit* u = &units[i]; const ctm_move_unit* nb = &units[u->nb];
            int y = chi;
            if (skip_zero_columns && u->n_rows >= chi && nb->n_rows >= chi) {
                const int ymax = std::max(u->ncol, nb->ncol);
                const int yc = std::min(chi, std::max(16, (ymax + 15) / 16 * 16));
                if (2 * yc <= chi) y = yc;
            }
            const int s = run_absorb(w, dir, u, nb, chi, y, normalize);
            if (s != CTM_OK) return s;
            if (hipStreamSynchronize(w->stream) != hipSuccess) { w->set_error("ctm_move: stream"); return (int)CTM_ERR_HIP; }
            return (int)CTM_OK;
        }, &err);
        if (st != CTM_OK) { drain(); if (!err.empty()) ctx->set_error("ctm_move, absorb unit: " + err); return st; }
        return CTM_OK;
    });
}
