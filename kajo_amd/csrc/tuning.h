// tuning.h -- environment overrides of launch-shaping constants, for the tools' library only.
//
// The product library (libkajo_hip.so) is built WITHOUT -DKAJO_TUNING: the macros below expand to nothing, the library
// contains no getenv and no knob names, and what bench.py times cannot be changed by a stray environment variable (the
// reference's only knobs are compile-time constants too: renderer/cpu/Shader.cpp:23-24, Renderer.cpp:21).
// `make -C kajo_amd/csrc tune` builds libkajo_hip_tune.so from the SAME kernel objects with capi.cpp / stage.cpp
// compiled -DKAJO_TUNING; tools/*.py select it with KAJO_HIP_LIB and say so in what they print.
#ifndef KAJO_TUNING_H
#define KAJO_TUNING_H

#ifdef KAJO_TUNING
#include <cstdlib>
// v = integer value of environment variable `name` when it is set and within [lo, hi]; else v is left alone
#define KAJO_TUNE_INT(name, lo, hi, v)                                                                                 \
    do {                                                                                                               \
        if (const char* e_ = std::getenv(name)) {                                                                      \
            const long x_ = std::atol(e_);                                                                             \
            if (x_ >= (long)(lo) && x_ <= (long)(hi))                                                                  \
                (v) = (decltype(v))x_;                                                                                 \
        }                                                                                                              \
    } while (0)
#define KAJO_TUNE_DOUBLE(name, v)                                                                                      \
    do {                                                                                                               \
        if (const char* e_ = std::getenv(name)) {                                                                      \
            const double x_ = std::atof(e_);                                                                           \
            if (x_ > 0)                                                                                                \
                (v) = x_;                                                                                              \
        }                                                                                                              \
    } while (0)
#define KAJO_TUNE_SET(name) (std::getenv(name) != nullptr)
#else
#define KAJO_TUNE_INT(name, lo, hi, v)                                                                                 \
    do {                                                                                                               \
    } while (0)
#define KAJO_TUNE_DOUBLE(name, v)                                                                                      \
    do {                                                                                                               \
    } while (0)
#define KAJO_TUNE_SET(name) (false)
#endif

#endif
