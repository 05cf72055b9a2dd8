#ifndef BPGPU_RANGEPROOF_H
#define BPGPU_RANGEPROOF_H
#include "msm_fixed.h"
#endif
