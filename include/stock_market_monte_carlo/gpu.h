// gpu.h -- drop-in for the reference's include/stock_market_monte_carlo/gpu.h:1-2, the vector-add demo
// north_star names beside the engine (src/gpu.cpp:7-15 on the host, src/gpu.cu:8-47 on the device).
#ifndef SMMC_DROPIN_GPU_H
#define SMMC_DROPIN_GPU_H

void vector_add(float *out, float *a, float *b, int n);      // host loop, prints "CPU time: <s>"
void vector_add_gpu(float *out, float *a, float *b, int n);  // MI355X (smmc_vector_add), prints "GPU time: <s>"

#endif
