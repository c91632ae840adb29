// index_params.h -- the constants that shape a mini-batch's word-major index: shared by the kernels
// that walk it (estep_kernels.h, where each one is explained) and by the host-only builder
// (batch_index.cpp), which cannot include HIP headers.
#pragma once

namespace trlda {

constexpr int kRegMaxN = 144;      // words of the longest register variant (8 waves x 18)
constexpr int kSplitMinN = 192;    // documents longer than this are split over several workgroups,
constexpr int kSplitSegN = 128;    // ceil(n / 128) segments of at most 128 words each,
constexpr int kSplitMaxSeg = 16;   // up to 16 of them (2048 words; longer documents: one workgroup)
constexpr int kLongWord = 16;
constexpr int kLongWordsTarget = 512;
constexpr int kSegMin = 256, kSegMax = 1024, kSegTasks = 512;
constexpr int kOneWaveMax = 256;             // long_len never exceeds this: a wave walks 16 entries per pass

}  // namespace trlda
