// libgla_hip.so -- the fp32 instances of the temporally blocked 5-point Jacobi pass (gla_march5.hip holds the kernel, the fp64
// instances and the entry point; the instances are split over two translation units because they take minutes to compile).
#define GLA_M5_F32_TU 1
#include "gla_march5.hip"
