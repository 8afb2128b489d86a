// ktx2.hpp -- a KTX 2.0 container to a sampled texture (vk_utils::create_ktx_texture,
// src/libs/vk_utils/vk_utils/tools.cpp:1043-1334; header and level index layouts :19-45).
//
// Like the reference: the 80-byte header is followed directly by the level index (one
// {byte_offset, byte_length, uncompressed_byte_length} per level, level 0 = the full image), a level's
// rows are tightly packed (bufferRowLength = level width, :1195-1197), stored levels are uploaded as
// they are, and level_count == 0 means "(uint32)log2(max(w, h)) levels, to be generated" (:1075-1078).
// The reference then still reads one index entry per *generated* level and never runs a blit
// (gen_mips is set and not used); here level 0 is taken from the file and the rest is built by the
// library's linear blits, which is what the flag announces.
// Supported: 2D, one layer, one face, no supercompression, 8-bit UNORM / SRGB formats with 1-4 channels.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "errors.hpp"
#include "handles.hpp"

namespace rnd_host {

struct ktx2_level_index {
    uint64_t byte_offset;
    uint64_t byte_length;
    uint64_t uncompressed_byte_length;
};

struct ktx2_header {
    unsigned char identifier[12];
    uint32_t vk_format;
    uint32_t type_size;
    uint32_t pixel_width;
    uint32_t pixel_height;
    uint32_t pixel_depth;
    uint32_t layer_count;
    uint32_t face_count;
    uint32_t level_count;
    uint32_t supercompression_scheme;
    uint32_t dfd_byte_offset;
    uint32_t dfd_byte_length;
    uint32_t kvd_byte_offset;
    uint32_t kvd_byte_length;
    uint64_t sgd_byte_offset;
    uint64_t sgd_byte_length;
};
static_assert(sizeof(ktx2_header) == 80, "KTX 2.0 header");

constexpr unsigned char ktx2_identifier[12] = {0xAB, 0x4B, 0x54, 0x58, 0x20, 0x32, 0x30, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A};

// bytes per texel of the formats the library samples, 0 otherwise
inline uint32_t ktx2_pixel_size(uint32_t vk_format)
{
    switch ((int32_t)vk_format) {
    case RND_FORMAT_R8_UNORM: case RND_FORMAT_R8_SRGB: return 1;
    case RND_FORMAT_R8G8_UNORM: case RND_FORMAT_R8G8_SRGB: return 2;
    case RND_FORMAT_R8G8B8_UNORM: case RND_FORMAT_R8G8B8_SRGB: return 3;
    case RND_FORMAT_R8G8B8A8_UNORM: case RND_FORMAT_R8G8B8A8_SRGB: return 4;
    default: return 0;
    }
}

struct ktx2_info {
    ktx2_header header{};
    std::vector<ktx2_level_index> levels; // as stored
    uint32_t texture_levels{0};           // levels the texture will have
    bool gen_mips{false};
};

inline ERROR_TYPE parse_ktx2(const void* data, size_t data_size, ktx2_info& out)
{
    if (data_size < sizeof(ktx2_header)) RAISE_ERROR_WARN(-1, "bad ktx file.");
    std::memcpy(&out.header, data, sizeof(ktx2_header));
    const ktx2_header& h = out.header;
    if (std::memcmp(h.identifier, ktx2_identifier, 12) != 0) RAISE_ERROR_WARN(-1, "bad ktx file."); // tools.cpp:1061-1063
    const uint32_t pixel_size = ktx2_pixel_size(h.vk_format);
    if (!pixel_size) RAISE_ERROR_WARN(-1, "unsupported vk format.");                               // tools.cpp:1065-1067
    if (h.pixel_width == 0 || h.pixel_height == 0 || h.pixel_depth != 0) RAISE_ERROR_WARN(-1, "only 2D ktx textures are supported.");
    if (h.layer_count > 1 || h.face_count != 1) RAISE_ERROR_WARN(-1, "array and cube ktx textures are not supported.");
    if (h.supercompression_scheme != 0) RAISE_ERROR_WARN(-1, "supercompressed ktx textures are not supported.");
    out.gen_mips = h.level_count == 0;
    const uint32_t stored = out.gen_mips ? 1u : h.level_count;
    if (out.gen_mips) {
        uint32_t m = std::max(h.pixel_width, h.pixel_height), l = 0;
        while ((2u << l) <= m) ++l; // (uint32_t)log2(max(w, h)), tools.cpp:1076
        out.texture_levels = std::max(l, 1u);
    } else {
        out.texture_levels = stored;
    }
    if (stored > 16) RAISE_ERROR_WARN(-1, "too many levels.");
    if (data_size < sizeof(ktx2_header) + (size_t)stored * sizeof(ktx2_level_index)) RAISE_ERROR_WARN(-1, "bad ktx file.");
    out.levels.resize(stored);
    std::memcpy(out.levels.data(), static_cast<const uint8_t*>(data) + sizeof(ktx2_header), (size_t)stored * sizeof(ktx2_level_index));
    uint32_t w = h.pixel_width, hh = h.pixel_height;
    for (uint32_t l = 0; l < stored; ++l) {
        const uint64_t need = (uint64_t)w * hh * pixel_size;
        const ktx2_level_index& li = out.levels[l];
        if (li.byte_offset > data_size || li.byte_length > data_size - li.byte_offset || li.byte_length < need)
            RAISE_ERROR_WARN(-1, "ktx level " + std::to_string(l) + " lies outside the file.");
        w = std::max(1u, w / 2u); hh = std::max(1u, hh / 2u); // tools.cpp:1218-1220
    }
    RAISE_ERROR_OK();
}

inline ERROR_TYPE create_ktx_texture(rnd_device dev, const void* data, size_t data_size, const rnd_sampler_info& sampler, texture_handler& out_texture)
{
    ktx2_info info;
    PASS_ERROR(parse_ktx2(data, data_size, info));
    const uint8_t* bytes = static_cast<const uint8_t*>(data);
    const ktx2_header& h = info.header;
    if (info.gen_mips) {
        RND_CHECK(rnd_texture_create_2d(dev, h.pixel_width, h.pixel_height, (int32_t)h.vk_format, 1, bytes + info.levels[0].byte_offset, &sampler, out_texture.put()));
    } else {
        std::vector<const void*> ptrs;
        for (const ktx2_level_index& li : info.levels) ptrs.push_back(bytes + li.byte_offset);
        RND_CHECK(rnd_texture_create_levels(dev, h.pixel_width, h.pixel_height, (int32_t)h.vk_format, (uint32_t)ptrs.size(), ptrs.data(), &sampler, out_texture.put()));
    }
    RAISE_ERROR_OK();
}

inline ERROR_TYPE read_file(const std::string& path, std::vector<uint8_t>& out)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) RAISE_ERROR_WARN(-1, "cannot open " + path);
    out.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
    RAISE_ERROR_OK();
}

} // namespace rnd_host
