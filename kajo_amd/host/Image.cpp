#include "Image.h"

#include <cstdio>
#include <iostream>
#include <vector>

Image::Image(int width, int height): width(width), height(height), pixels(new uint32_t[(size_t)width * height]())
{
}

namespace
{

uint32_t crc32(const unsigned char* p, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++)
                c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++)
        crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}

void be32(std::vector<unsigned char>& v, uint32_t x)
{
    v.push_back(x >> 24);
    v.push_back(x >> 16);
    v.push_back(x >> 8);
    v.push_back(x);
}

void chunk(std::vector<unsigned char>& out, const char* type, const std::vector<unsigned char>& data)
{
    be32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    be32(out, crc32(&out[start], out.size() - start));
}

} // namespace

bool Image::save(const std::string& fileName) const
{
    // raw scanlines: filter byte 0 + RGBA (the reference swaps R and B for its encoder the same way,
    // renderer/Image.cpp:34-38)
    std::vector<unsigned char> raw;
    raw.reserve((size_t)height * (1 + 4 * (size_t)width));
    for (int y = 0; y < height; y++) {
        raw.push_back(0);
        for (int x = 0; x < width; x++) {
            uint32_t p = pixels[(size_t)y * width + x];
            raw.push_back((p >> 16) & 0xff);
            raw.push_back((p >> 8) & 0xff);
            raw.push_back(p & 0xff);
            raw.push_back((p >> 24) & 0xff);
        }
    }
    // zlib stream of stored blocks
    std::vector<unsigned char> z;
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (unsigned char c : raw) {
        a = (a + c) % 65521u;
        b = (b + a) % 65521u;
    }
    for (size_t pos = 0; pos < raw.size() || pos == 0; pos += 65535) {
        size_t len = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + len >= raw.size() ? 1 : 0);
        z.push_back(len & 0xff);
        z.push_back(len >> 8);
        z.push_back(~len & 0xff);
        z.push_back((~len >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + len);
        if (raw.empty())
            break;
    }
    be32(z, (b << 16) | a);

    std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> ihdr;
    be32(ihdr, (uint32_t)width);
    be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); // bit depth
    ihdr.push_back(6); // RGBA
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    chunk(png, "IHDR", ihdr);
    chunk(png, "IDAT", z);
    chunk(png, "IEND", {});

    FILE* f = std::fopen(fileName.c_str(), "wb");
    if (!f) {
        std::cerr << "PNG encode failure: cannot open " << fileName << std::endl;
        return false;
    }
    bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    ok = std::fclose(f) == 0 && ok;
    return ok;
}
