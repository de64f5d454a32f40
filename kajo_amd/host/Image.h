// Image.h -- stand-in for the frame object a Kajo backend fills, so that hip::Scheduler builds without the
// Kajo tree. A backend touches exactly three members of the reference's class (renderer/Image.h:9-21), by name:
//   width, height   frame size in pixels
//   pixels          width * height words, 0xAARRGGBB, sRGB-encoded, top row first
// The colour conversion helpers of the reference class are not needed here: the resolve runs on the GPU
// (kajo_hip_resolve_argb8).
#ifndef KAJO_HOST_IMAGE_H
#define KAJO_HOST_IMAGE_H

#include <cstdint>
#include <memory>
#include <string>

class Image
{
public:
    int width = 0;
    int height = 0;
    std::unique_ptr<uint32_t[]> pixels;

    Image(int w, int h);

    // Writes an 8-bit RGBA PNG whose scanlines sit in stored (uncompressed) deflate blocks; false on I/O failure.
    bool save(const std::string& path) const;
};

#endif
