// Device tensors + pairwise-einsum executor (contract.hip).
#pragma once
#include "ctm_common.h"

struct DT {
    double* p = nullptr;          // the tensor (real contexts) or its real plane
    double* q = nullptr;          // imaginary plane (complex128 contexts), same layout as p
    bool cj = false;              // operand is read conjugated (folded into the GEMM signs, never materialised)
    std::vector<long long> dims;
    DT() {}
    DT(const double* ptr, std::initializer_list<long long> d) : p(const_cast<double*>(ptr)), dims(d) {}
    DT(const double* ptr, const std::vector<long long>& d) : p(const_cast<double*>(ptr)), dims(d) {}
    long long numel() const;
    DT view(std::initializer_list<long long> d) const { DT t = *this; t.dims = d; return t; }
    DT view(const std::vector<long long>& d) const { DT t = *this; t.dims = d; return t; }
    DT conj() const { DT t = *this; if (q) t.cj = !cj; return t; }
};

// C[io] = sum over indices shared by ia, ib and absent from io.  If out->p != nullptr the result is
// written there, otherwise it is arena-allocated (valid until the caller's ArenaScope ends).
int dev_einsum2(ctm_ctx* ctx, const std::string& ia, const DT& A, const std::string& ib, const DT& B,
                const std::string& io, DT* out);
// "ab,bcd,...->xyz" evaluated strictly left to right (same order as the oracle's seq_einsum).
int dev_seq_einsum(ctm_ctx* ctx, const std::string& expr, const std::vector<DT>& ops, DT* out);

// Fused two-layer site absorption (layer2.hip): out[io] = sum Z[iz] a[ia] a[ib] with both site operands the same tensor.
bool layer2_roles(const std::string& iz, const DT& Z, const std::string& ia, const std::string& ib, const DT& A,
                  char ck[2], char cb[2], char ek[2], char eb[2], char sp[2]);
int dev_layer2(ctm_ctx* ctx, const std::string& iz, const DT& Z, const std::string& ia, const std::string& ib, const DT& A,
               const std::string& io, DT* out);
// Whole network "i0,i1,...->o": like dev_seq_einsum, but a consecutive (a, conj a) pair of site operands is absorbed by
// the fused kernel when the pattern fits (otherwise plain left-to-right pairwise evaluation).
int dev_network(ctm_ctx* ctx, const std::string& expr, const std::vector<DT>& ops, DT* out);
