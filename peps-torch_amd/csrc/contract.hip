// Device tensor-network contraction executor: pairwise einsum -> (optional coalesced permute) + FP64
// MFMA GEMM.  Replaces the reference's tn_interface.contract/einsum (tn_interface.py:3-27), i.e. every
// torch.tensordot / torch.einsum on the hot path.  Index strings are single letters; a pairwise step
// has no batch (Hadamard) indices in any CTM contraction, so it is exactly one GEMM:
//   C[freeA..., freeB...] = sum_K A[...] B[...]
// An operand is used in place (N or T form, arbitrary leading stride) whenever its contracted indices
// form a contiguous prefix/suffix in memory in a K-order both operands agree on; otherwise the SMALLER
// operand is re-laid-out by the tiled permute kernel.
#include "contract.h"
#include <algorithm>

long long DT::numel() const { long long n = 1; for (auto d : dims) n *= d; return n; }

namespace {

bool is_prefix(const std::string& idx, const std::string& c) { return idx.compare(0, c.size(), c) == 0; }
bool is_suffix(const std::string& idx, const std::string& c) {
    return idx.size() >= c.size() && idx.compare(idx.size() - c.size(), c.size(), c) == 0;
}

long long dim_of(const std::string& idx, const DT& t, char ch) {
    const size_t p = idx.find(ch);
    return t.dims[p];
}

// permute tensor t (indices `from`) into index order `to` (arena-allocated)
int relayout(ctm_ctx* ctx, const std::string& from, const DT& t, const std::string& to, DT* out) {
    const int nd = (int)from.size();
    long long dims[CTM_MAXD]; int perm[CTM_MAXD];
    out->dims.resize(nd);
    for (int a = 0; a < nd; ++a) {
        dims[a] = t.dims[a];
        perm[a] = (int)from.find(to[a]);
        out->dims[a] = t.dims[perm[a]];
    }
    const size_t ne = (size_t)t.numel();
    if (!out->p) {
        CTM_TRY(arena_alloc(ctx, sizeof(double) * ne * (t.q ? 2 : 1), (void**)&out->p));
        if (t.q) out->q = out->p + ne;
    }
    out->cj = t.cj;
    CTM_TRY(permute_f64(ctx, t.p, out->p, nd, dims, perm));
    if (t.q) CTM_TRY(permute_f64(ctx, t.q, out->q, nd, dims, perm));
    return CTM_OK;
}

}  // namespace

int dev_einsum2(ctm_ctx* ctx, const std::string& ia_, const DT& A_, const std::string& ib_, const DT& B_,
                const std::string& io, DT* out) {
    std::string ia = ia_, ib = ib_;
    DT A = A_, B = B_;
    if ((int)ia.size() != (int)A.dims.size() || (int)ib.size() != (int)B.dims.size()) {
        ctx->set_error("einsum2: rank mismatch " + ia + "," + ib); return CTM_ERR_SHAPE;
    }
    std::string cA, cB, fA, fB;
    for (char ch : ia) { if (ib.find(ch) != std::string::npos && io.find(ch) == std::string::npos) cA += ch; else fA += ch; }
    for (char ch : ib) { if (ia.find(ch) != std::string::npos && io.find(ch) == std::string::npos) cB += ch; else fB += ch; }
    for (char ch : cA) if (dim_of(ia, A, ch) != dim_of(ib, B, ch)) {
        ctx->set_error(std::string("einsum2: dim mismatch on index ") + ch + " in " + ia + "," + ib); return CTM_ERR_SHAPE;
    }
    for (char ch : fA) if (fB.find(ch) != std::string::npos) { ctx->set_error("einsum2: batch index unsupported"); return CTM_ERR_UNSUPPORTED; }
    if (cA.empty()) { ctx->set_error("einsum2: outer product unsupported"); return CTM_ERR_UNSUPPORTED; }

    // The requested output order is the natural one (free of A, free of B) up to a permutation INSIDE each group, and the
    // result is bigger than the operands: permute the (small) operands so that the GEMM writes the requested layout directly
    // instead of permuting the (large) result afterwards.
    if (ctx->einsum_in_relayout && io != fA + fB && io.size() == fA.size() + fB.size()) {
        const std::string ga = io.substr(0, fA.size()), gb = io.substr(fA.size());
        auto is_perm = [](std::string a, std::string b) { std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); return a == b; };
        long long mn = 1;
        for (char ch : fA) mn *= dim_of(ia, A, ch);
        for (char ch : fB) mn *= dim_of(ib, B, ch);
        if (is_perm(ga, fA) && is_perm(gb, fB) && A.numel() + B.numel() <= mn) {
            const bool a_fits = (ia == ga + cA || ia == cA + ga), b_fits_a = (ib == cA + gb || ib == gb + cA);
            const bool a_fits_b = (ia == ga + cB || ia == cB + ga), b_fits = (ib == cB + gb || ib == gb + cB);
            // contraction-index order: the one that leaves the larger operand untouched if possible
            std::string kord = cA;
            if (!(a_fits && b_fits_a) && ((a_fits_b && b_fits) || (!a_fits && b_fits && B.numel() >= A.numel()))) kord = cB;
            if (!(ia == ga + kord || ia == kord + ga)) { DT An; const std::string to = ga + kord; CTM_TRY(relayout(ctx, ia, A, to, &An)); A = An; ia = to; }
            if (!(ib == kord + gb || ib == gb + kord)) { DT Bn; const std::string to = kord + gb; CTM_TRY(relayout(ctx, ib, B, to, &Bn)); B = Bn; ib = to; }
            cA = kord; cB = kord; fA = ga; fB = gb;
        }
    }

    bool A_ok = is_prefix(ia, cA) || is_suffix(ia, cA);
    bool B_ok = is_prefix(ib, cB) || is_suffix(ib, cB);
    std::string korder;
    if (A_ok && B_ok && cA == cB) korder = cA;
    else if (A_ok && B_ok) { if (A.numel() >= B.numel()) { korder = cA; B_ok = false; } else { korder = cB; A_ok = false; } }
    else if (A_ok) korder = cA;
    else if (B_ok) korder = cB;
    else korder = (A.numel() >= B.numel()) ? cA : cB;
    if (!A_ok || (korder != cA)) {
        if (!(is_prefix(ia, korder) || is_suffix(ia, korder))) {
            DT An; const std::string to = fA + korder;
            CTM_TRY(relayout(ctx, ia, A, to, &An)); A = An; ia = to;
        }
    }
    if (!B_ok || (korder != cB)) {
        if (!(is_prefix(ib, korder) || is_suffix(ib, korder))) {
            DT Bn; const std::string to = korder + fB;
            CTM_TRY(relayout(ctx, ib, B, to, &Bn)); B = Bn; ib = to;
        }
    }
    long long M = 1, N = 1, K = 1;
    for (char ch : fA) M *= dim_of(ia, A, ch);
    for (char ch : fB) N *= dim_of(ib, B, ch);
    for (char ch : korder) K *= dim_of(ia, A, ch);
    const bool cx = (A.q != nullptr) || (B.q != nullptr);
    if (cx && !(A.q && B.q)) { ctx->set_error("einsum2: mixed real/complex operands"); return CTM_ERR_BADARG; }
    XM xa, xb;
    xa.re = A.p; xa.im = A.q; xa.c = A.cj; if (is_suffix(ia, korder)) { xa.ld = K; xa.t = false; } else { xa.ld = M; xa.t = true; }
    xb.re = B.p; xb.im = B.q; xb.c = B.cj; if (is_prefix(ib, korder)) { xb.ld = N; xb.t = false; } else { xb.ld = K; xb.t = true; }
    const std::string nat = fA + fB;
    DT C;
    C.dims.clear();
    for (char ch : fA) C.dims.push_back(dim_of(ia, A, ch));
    for (char ch : fB) C.dims.push_back(dim_of(ib, B, ch));
    const bool direct = (nat == io);
    if (direct && out->p) { C.p = out->p; C.q = out->q; }
    else {
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)(M * N) * (cx ? 2 : 1), (void**)&C.p));
        if (cx) C.q = C.p + M * N;
    }
    if (cx && !C.q) { ctx->set_error("einsum2: complex result needs two planes"); return CTM_ERR_BADARG; }
    CTM_TRY(xgemm(ctx, (int)M, (int)N, (int)K, xa, xb, C.p, C.q, N));
    if (direct) { out->p = C.p; out->q = C.q; out->cj = false; out->dims = C.dims; return CTM_OK; }
    return relayout(ctx, nat, C, io, out);
}

int dev_seq_einsum(ctm_ctx* ctx, const std::string& expr, const std::vector<DT>& ops, DT* out) {
    const size_t arrow = expr.find("->");
    const std::string lhs = expr.substr(0, arrow), oidx = expr.substr(arrow + 2);
    std::vector<std::string> ins;
    { size_t s = 0; while (true) { size_t c = lhs.find(',', s); ins.push_back(lhs.substr(s, c == std::string::npos ? c : c - s)); if (c == std::string::npos) break; s = c + 1; } }
    if (ins.size() != ops.size() || ins.size() < 2) { ctx->set_error("seq_einsum: operand count"); return CTM_ERR_BADARG; }
    DT cur = ops[0];
    std::string cidx = ins[0];
    for (size_t k = 1; k < ins.size(); ++k) {
        const bool last = (k + 1 == ins.size());
        std::string later = oidx;
        for (size_t j = k + 1; j < ins.size(); ++j) later += ins[j];
        std::string nidx;
        for (char ch : cidx + ins[k]) if (later.find(ch) != std::string::npos && nidx.find(ch) == std::string::npos) nidx += ch;
        DT nxt;
        if (last) { nxt.p = out->p; nxt.q = out->q; CTM_TRY(dev_einsum2(ctx, cidx, cur, ins[k], ops[k], oidx, &nxt)); }
        else CTM_TRY(dev_einsum2(ctx, cidx, cur, ins[k], ops[k], nidx, &nxt));
        cur = nxt; cidx = last ? oidx : nidx;
    }
    *out = cur;
    return CTM_OK;
}

namespace {
std::vector<std::string> split_inputs(const std::string& lhs) {
    std::vector<std::string> ins; size_t s = 0;
    while (true) { size_t c = lhs.find(',', s); ins.push_back(lhs.substr(s, c == std::string::npos ? c : c - s)); if (c == std::string::npos) break; s = c + 1; }
    return ins;
}
std::string join_expr(const std::vector<std::string>& ins, size_t a, size_t b, const std::string& o) {
    std::string e;
    for (size_t i = a; i < b; ++i) { if (i > a) e += ","; e += ins[i]; }
    return e + "->" + o;
}
}  // namespace

int dev_network(ctm_ctx* ctx, const std::string& expr, const std::vector<DT>& ops, DT* out) {
    const size_t arrow = expr.find("->");
    const std::string oidx = expr.substr(arrow + 2);
    const std::vector<std::string> ins = split_inputs(expr.substr(0, arrow));
    size_t k = 0;
    bool found = false;
    if (ctx->use_layer2 && (!ops[0].q || ctx->layer2_cplx))
        for (k = 1; k + 1 < ins.size(); ++k)
            if (ops[k].p == ops[k + 1].p && ins[k].size() == 5 && ins[k + 1].size() == 5 && ins[k][0] == ins[k + 1][0] &&
                (!ops[k].q || (!ops[k].cj && ops[k + 1].cj))) { found = true; break; }      // complex: (a, conj a) in this order
    if (!found) return dev_seq_einsum(ctx, expr, ops, out);
    // natural index order of the prefix result (same rule as dev_seq_einsum)
    std::string later = oidx;
    for (size_t j = k; j < ins.size(); ++j) later += ins[j];
    std::string zidx = ins[0];
    for (size_t i = 1; i < k; ++i) {
        std::string need = oidx;
        for (size_t j = i + 1; j < ins.size(); ++j) need += ins[j];
        std::string nidx;
        for (char ch : zidx + ins[i]) if (need.find(ch) != std::string::npos && nidx.find(ch) == std::string::npos) nidx += ch;
        zidx = nidx;
    }
    // The fused kernel gathers, per spectator pair (x,y), the block of the four site-contracted indices: it wants those to be
    // the fast-running ones.  If the natural order ends on a spectator index (e.g. LD corner "xlL"+"dDy", every absorb
    // "bcd"+"ije"), move the spectators of the last operand's group to the front of that group -- the GEMM that produces Z
    // then writes 64-double runs per (x,y) instead of one element per cache line (dev_einsum2 permutes the small operand).
    if (ctx->z_spectators_first && k >= 2 && zidx.size() == 6) {
        auto is_site = [&](char ch) { return ins[k].find(ch) != std::string::npos || ins[k + 1].find(ch) != std::string::npos; };
        if (!is_site(zidx.back())) {
            size_t g0 = zidx.size();                   // start of the group contributed by the last prefix operand
            while (g0 > 0 && ins[k - 1].find(zidx[g0 - 1]) != std::string::npos) --g0;
            std::string head = zidx.substr(0, g0), spect, site;
            for (char ch : zidx.substr(g0)) (is_site(ch) ? site : spect) += ch;
            if (!site.empty()) zidx = head + spect + site;
        }
    }
    DT Z;
    if (k == 1) Z = ops[0];
    else {
        std::vector<DT> pre(ops.begin(), ops.begin() + k);
        CTM_TRY(dev_seq_einsum(ctx, join_expr(ins, 0, k, zidx), pre, &Z));
    }
    char ck[2], cb[2], ek[2], eb[2], sp[2];
    if (zidx.size() != 6 || !layer2_roles(zidx, Z, ins[k], ins[k + 1], ops[k], ck, cb, ek, eb, sp)) {
        // pattern does not fit: finish pairwise from Z
        std::vector<DT> rest; rest.push_back(Z);
        std::vector<std::string> rins; rins.push_back(zidx);
        for (size_t j = k; j < ins.size(); ++j) { rest.push_back(ops[j]); rins.push_back(ins[j]); }
        return dev_seq_einsum(ctx, join_expr(rins, 0, rins.size(), oidx), rest, out);
    }
    const std::string six = std::string() + sp[0] + sp[1] + ek[0] + eb[0] + ek[1] + eb[1];
    const bool has_suffix = (k + 2 < ins.size());
    std::string io;
    if (!has_suffix) {
        io = oidx;
        if (io.size() != 6) { ctx->set_error("network: fused output rank"); return CTM_ERR_BADARG; }
    } else {
        const std::string& nx = ins[k + 2];
        std::string keep_later = oidx;
        for (size_t j = k + 3; j < ins.size(); ++j) keep_later += ins[j];
        std::string cn, others;
        for (char ch : nx) if (six.find(ch) != std::string::npos && keep_later.find(ch) == std::string::npos) cn += ch;
        for (char ch : oidx) if (six.find(ch) != std::string::npos && cn.find(ch) == std::string::npos) others += ch;
        for (char ch : six) if (cn.find(ch) == std::string::npos && others.find(ch) == std::string::npos) others += ch;
        io = others + cn;
    }
    DT O;
    long long numel = 1;
    {   // dims of the six output letters
        auto dimA = [&](char ch) { return ops[k].dims[ins[k].find(ch)]; };
        for (char ch : io) {
            long long d;
            if (ch == sp[0] || ch == sp[1]) d = Z.dims[zidx.find(ch)];
            else if (ch == ek[0] || ch == eb[0]) d = dimA(ek[0]);
            else d = dimA(ek[1]);
            numel *= d;
        }
    }
    const bool cx = ops[k].q != nullptr;
    if (!has_suffix && out->p) { O.p = out->p; O.q = out->q; }
    else {
        CTM_TRY(arena_alloc(ctx, sizeof(double) * (size_t)numel * (cx ? 2 : 1), (void**)&O.p));
        if (cx) O.q = O.p + numel;
    }
    CTM_TRY(dev_layer2(ctx, zidx, Z, ins[k], ins[k + 1], ops[k], io, &O));
    if (!has_suffix) { *out = O; return CTM_OK; }
    std::vector<DT> rest; rest.push_back(O);
    std::vector<std::string> rins; rins.push_back(io);
    for (size_t j = k + 2; j < ins.size(); ++j) { rest.push_back(ops[j]); rins.push_back(ins[j]); }
    return dev_seq_einsum(ctx, join_expr(rins, 0, rins.size(), oidx), rest, out);
}
