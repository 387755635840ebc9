// png_writer.hpp -- minimal 8-bit RGBA PNG encoder (zlib deflate), the stand-in for the reference's
// stbi_write_png(file, w, h, 4, image, w*4) call (volumetric-ray-tracer/main.cpp:306).
// Like the reference it is handed the u32 framebuffer as BYTES: A<<24|R<<16|G<<8|B is stored
// little-endian, so the file's "R" channel holds blue and its "B" channel holds red -- the reference's
// files have the same swap and tools that read them back rely on it.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <vector>

namespace png {

inline void put_be32(std::vector<unsigned char> &v, uint32_t x)
{
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
inline void chunk(FILE *f, const char type[4], const std::vector<unsigned char> &data)
{
    std::vector<unsigned char> head;
    put_be32(head, (uint32_t)data.size());
    fwrite(head.data(), 1, 4, f);
    uLong crc = crc32(0L, (const Bytef *)type, 4);
    if (!data.empty()) crc = crc32(crc, data.data(), (uInt)data.size());
    fwrite(type, 1, 4, f);
    if (!data.empty()) fwrite(data.data(), 1, data.size(), f);
    std::vector<unsigned char> tail;
    put_be32(tail, (uint32_t)crc);
    fwrite(tail.data(), 1, 4, f);
}

// returns true on success
inline bool write_rgba(const char *path, uint32_t w, uint32_t h, const void *pixels, size_t stride_bytes)
{
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    fwrite(sig, 1, 8, f);
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, w); put_be32(ihdr, h);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); // 8-bit RGBA
    chunk(f, "IHDR", ihdr);
    std::vector<unsigned char> raw((size_t)h * (1 + (size_t)w * 4));
    for (uint32_t y = 0; y < h; ++y) {
        unsigned char *row = raw.data() + (size_t)y * (1 + (size_t)w * 4);
        row[0] = 0; // filter: none
        const unsigned char *src = (const unsigned char *)pixels + (size_t)y * stride_bytes;
        for (size_t i = 0; i < (size_t)w * 4; ++i) row[1 + i] = src[i];
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { fclose(f); return false; }
    z.resize(zlen);
    chunk(f, "IDAT", z);
    chunk(f, "IEND", {});
    return fclose(f) == 0;
}

} // namespace png
