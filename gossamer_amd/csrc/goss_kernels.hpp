// goss_kernels.hpp -- hand-written HIP kernels for gfx950 (MI355X): key extraction,
// LSD radix sort, run compaction, Elias-Fano (SparseArray) + DenseSelect image build.
// All integer / byte work, HBM-bound; no MFMA by design.
//
// Wave size is 64 everywhere; workgroups are 256 threads (4 waves, one per SIMD).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "goss_key.hpp"

#include "kernels_common.hpp"
#include "kernels_extract.hpp"
#include "kernels_route.hpp"
#include "kernels_partition.hpp"
#include "kernels_runs.hpp"
#include "kernels_text.hpp"
#include "kernels_count.hpp"
#include "kernels_merge.hpp"
#include "kernels_emit.hpp"
