// Image.h -- the output image a backend fills (renderer/Image.h:9-21 in the reference):
// width x height uint32 ARGB8 (A << 24 | R << 16 | G << 8 | B), sRGB-encoded, row 0 = top.
#ifndef KAJO_HOST_IMAGE_H
#define KAJO_HOST_IMAGE_H

#include <cstdint>
#include <memory>
#include <string>

class Image
{
public:
    Image(int width, int height);
    // PNG (8-bit RGBA). The encoder stores the scanlines in uncompressed deflate blocks.
    bool save(const std::string& fileName) const;

    int width;
    int height;
    std::unique_ptr<uint32_t[]> pixels;
};

#endif
