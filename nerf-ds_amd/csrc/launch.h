// Launch-side helpers shared by the three fused-kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#ifndef NERFDS_GRAPH
#error "compile with -DNERFDS_GRAPH=<GraphNerfDS|GraphStatic|GraphHyperNeRF> -DNERFDS_NAME=<suffix> (render: -DNERFDS_PREC=<P_BF16|P_BF16X3|P_F32|P_F16> or -DNERFDS_MIXED)"
#endif
#define NERFDS_CAT2(a, b) a##b
#define NERFDS_CAT(a, b) NERFDS_CAT2(a, b)

#include "lds_attr.h"   // allow_dynamic_lds: the > 64 KiB dynamic-LDS attribute, once per (kernel, device)
