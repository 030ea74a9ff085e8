// image_decode.h — PNG / JPEG bytes to RGBA8 (what stbi_load_from_memory(..., 4) gives the reference).
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace app {

struct DecodedImage {
    int width = 0, height = 0;
    int channels = 0;            // channels of the file (the pixels below are always RGBA)
    std::vector<uint8_t> rgba;   // rows top to bottom
};

// Throws std::runtime_error("Unable to load image from memory: ...") on anything it cannot decode.
DecodedImage decode_image(const uint8_t* data, size_t size);

}  // namespace app
