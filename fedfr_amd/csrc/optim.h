// flat-buffer optimiser / aggregation / PartialFC sampling kernels (see optim.hip)
#pragma once
#include "common.h"

int optim_sgd(float* p, float* g, float* buf, bf16_t* shadow, size_t n, float lr, float mu, float wd, int first,
              hipStream_t st, float gscale = 1.f, unsigned* overflow = nullptr);   // gscale != 1: g holds gradient / gscale; it is multiplied in the kernel and stored back
int optim_fedavg_axpy(float* dst, const float* src, float w, size_t n, int accumulate, hipStream_t st);
int optim_fedavg_multi(float* dst, const float* const* srcs, const float* ws, int k, size_t n, int accumulate, hipStream_t st);
int optim_fedavg_i64(float* acc, const long long* src, float w, int n, int accumulate, long long* out_trunc, hipStream_t st);
int optim_pfc_rand(float* perm, int n, unsigned long long seed, unsigned long long step, hipStream_t st);
int optim_pfc_localize(long long* label, int n, long long class_start, int num_local, float* perm, hipStream_t st);
int optim_pfc_topk(const float* perm, int n, int k, long long* index, int* npos_out, hipStream_t st);
int optim_pfc_positive(const float* perm, int n, long long* index, int* count, hipStream_t st);
int optim_pfc_remap(long long* label, int n, const long long* index, int k, hipStream_t st);
int optim_rows(float* dst, const float* src, const long long* index, int k, int D, int scatter, int nrows, hipStream_t st);
