// vkr_zstd.h — Zstandard frame decoder of the bake library (vkr_zstd.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

// true if the buffer starts with a Zstandard (or skippable) frame magic
bool vkr_zstd_is_frame(const uint8_t* src, size_t len);
// Appends the content of every frame in [src, src + len) to `out`.  false + `error` on malformed input.
bool vkr_zstd_decompress(const uint8_t* src, size_t len, std::vector<uint8_t>& out, std::string& error);
