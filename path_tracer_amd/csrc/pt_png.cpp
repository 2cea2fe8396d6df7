#include "pt_png.h"

#include <cstdio>
#include <vector>

namespace pt {
namespace {
uint32_t crc_table[256];
bool crc_ready = false;
void crc_init()
{
    for (uint32_t n = 0; n < 256; ++n)
    {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
    crc_ready = true;
}
uint32_t crc32(const uint8_t* p, size_t n, uint32_t c = 0xFFFFFFFFu)
{
    for (size_t i = 0; i < n; ++i) c = crc_table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return c;
}
void be32(std::vector<uint8_t>& v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
bool chunk(FILE* f, const char type[4], const std::vector<uint8_t>& data)
{
    std::vector<uint8_t> head;
    be32(head, (uint32_t)data.size());
    head.insert(head.end(), type, type + 4);
    uint32_t c = crc32(head.data() + 4, 4);
    c = crc32(data.data(), data.size(), c) ^ 0xFFFFFFFFu;
    std::vector<uint8_t> tail;
    be32(tail, c);
    return fwrite(head.data(), 1, 8, f) == 8 && (data.empty() || fwrite(data.data(), 1, data.size(), f) == data.size()) &&
           fwrite(tail.data(), 1, 4, f) == 4;
}
} // namespace

bool write_png_rgb8(const char* path, const uint8_t* rgb, uint32_t w, uint32_t h, std::string* err)
{
    if (!crc_ready) crc_init();
    const size_t row = (size_t)w * 3u, raw_size = (row + 1u) * h;
    if (raw_size > 0xF0000000ull) { if (err) *err = "image too large for a single IDAT chunk"; return false; }
    std::vector<uint8_t> raw(raw_size);
    for (uint32_t y = 0; y < h; ++y)
    {
        raw[(row + 1u) * y] = 0; // filter: none
        for (size_t i = 0; i < row; ++i) raw[(row + 1u) * y + 1u + i] = rgb[row * y + i];
    }
    std::vector<uint8_t> z;
    z.reserve(raw_size + raw_size / 65535u * 5u + 16u);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0; // adler32
    size_t pos = 0;
    do
    {
        const size_t n = raw_size - pos < 65535u ? raw_size - pos : 65535u;
        z.push_back(pos + n == raw_size ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        for (size_t i = 0; i < n; ++i)
        {
            const uint8_t v = raw[pos + i];
            z.push_back(v);
            a += v; if (a >= 65521u) a -= 65521u;
            b += a; if (b >= 65521u) b -= 65521u;
        }
        pos += n;
    } while (pos < raw_size);
    be32(z, (b << 16) | a);

    FILE* f = fopen(path, "wb");
    if (!f) { if (err) *err = std::string("cannot open ") + path + " for writing"; return false; }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, w); be32(ihdr, h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    bool ok = fwrite(sig, 1, 8, f) == 8 && chunk(f, "IHDR", ihdr) && chunk(f, "IDAT", z) && chunk(f, "IEND", {});
    ok = (fclose(f) == 0) && ok;
    if (!ok && err) *err = std::string("short write to ") + path;
    return ok;
}

} // namespace pt
