// Internal (not exported): the B-resident skinny GEMM of gemm_skinny.hip, dispatched from renet_gemm_f32_split.
#pragma once
bool renet_gemm_skinny_eligible(int ta, int M, int N, int K, const float* A, int lda, const float* B, int ldb, int tb);
int renet_gemm_skinny_launch(int tb, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
                             float beta, float* C, int ldc, const float* bias, void* stream);
