// Minimal PNG encoder for ImageHelper::write_image (src/image_helper.rs:50-57: image::save_buffer(path, data, w, h, Rgb8)).
// 8-bit RGB, filter 0 on every row, zlib stream of stored (uncompressed) deflate blocks: any decoder reads it, the pixels
// are what matters.  Host only.
#pragma once
#include <cstdint>
#include <string>

namespace pt {
bool write_png_rgb8(const char* path, const uint8_t* rgb, uint32_t w, uint32_t h, std::string* err);
}
