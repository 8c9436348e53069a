// fp32 head kernels (see head.hip).
#pragma once
#include "common.h"

int head_normalize_rows(const float* x, float* xn, float* inv, int R, int D, float eps, hipStream_t st);
int head_normalize_rows_bwd(const float* xn, const float* inv, const float* dxn, float* dx, int R, int D, float beta,
                            hipStream_t st);
int head_sgemm(const float* A, const float* B, float* C, int M, int N, int K, long long sam, long long sak,
               long long sbk, long long sbn, int ldc, float alpha, float beta, const float* bias, hipStream_t st);
int head_normalize_rows_bwd_slabs(const float* xn, const float* inv, const float* dxn, int nslab, long long slab_stride, float* dx, int R, int D,
                                  float beta, hipStream_t st);
int head_sgemm_splitk(const float* A, const float* B, float* C, int M, int N, int K, long long sam, long long sak, long long sbk, long long sbn,
                      int ldc, float alpha, int splits, long long slab_stride, hipStream_t st);
int head_softmax_ce_fused(float* z, const long long* label, int R, int C, int ldz, float s, float m, int arc, float inv_batch, float* prob_t,
                          int nslab, long long slab_stride, hipStream_t st);
// same, products accumulated in fp64 (fp32 validation path of the backbone)
int head_sgemm_f64acc(const float* A, const float* B, float* C, int M, int N, int K, long long sam, long long sak, long long sbk, long long sbn,
                      int ldc, float alpha, float beta, const float* bias, hipStream_t st);
int head_margin_rowmax(float* z, const long long* label, int R, int C, int ldz, float s, float m, int arc, float* row_max,
                       float* dmul, hipStream_t st);
int head_exp_rowsum(float* z, int R, int C, int ldz, const float* row_max, float* row_sum, hipStream_t st);
int head_softmax_grad(float* z, const long long* label, int R, int C, int ldz, const float* row_sum, const float* dmul, float s,
                      float inv_batch, float* prob_t, hipStream_t st);
int head_margin_bwd(const float* dlogits, const long long* label, const float* dmul, float s, int R, int C, float* dcos,
                    hipStream_t st);
int head_nll_mean(const float* prob_t, int R, float floor_, float* loss, hipStream_t st);
int head_exp_rowsum_target(float* z, const long long* label, int R, int C, int ldz, const float* row_max, float* sums2, hipStream_t st);
int head_nll_mean_ratio(const float* num, const float* den, int R, float floor_, float* loss, hipStream_t st);
int head_bce_logits(const float* cosv, const long long* label, const float* bias, int B, int C, float m, float r, float t,
                    float* z, unsigned char* gt, float* dzdcos, hipStream_t st);
int head_bce_loss(const float* z, const unsigned char* gt, const float* dzdcos, int B, int C, float r, float lam, float loss_scale,
                  float* dz, float* dcos, float* row_loss, hipStream_t st);
int head_colsum_f32(const float* x, int R, int C, float* out, hipStream_t st);
int head_sum_scale(const float* x, int n, float scale, float* out, hipStream_t st);
int head_contrastive(const float* x, const float* g, const float* l, int B, int D, float temperature, float* row_loss, float* dx,
                     hipStream_t st);
int head_sgemm_colflag(const float* A, const float* B, int M, int N, int K, long long sam, long long sak, long long sbk,
                       long long sbn, float alpha, float thr, unsigned char* flags, hipStream_t st);
int head_class_accumulate(const float* x, const long long* label, int B, int D, int C, float* sums, float* counts, hipStream_t st);
int head_roc_histogram(const float* feat, const long long* label, int N, int D, int T, unsigned long long* hist, hipStream_t st);
