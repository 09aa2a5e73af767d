// CPU-side image store with the reference's Framebuffer layout (Framebuffer.h:12-58 in the reference):
//   colorBuffer         width*height RGBA8 display pixels
//   accumulationBuffer  width*height*3 floats, zero-initialised linear radiance
// The HIP kernels accumulate in device memory; this struct is where a host reads the result back to, in the
// layout existing code expects.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

// clamp to [0, 1]; a NaN becomes 0 (the reference's form, f < 0 ? 0 : (f > 1 ? 1 : f), hands a NaN on to a float -> uint8
// conversion, which is undefined)
inline float saturate(float f) { return !(f > 0.f) ? 0.f : (f > 1.f ? 1.f : f); }

struct Color {
    uint8_t R, G, B, A;

    // Display transfer of the reference's CPU path: square root, then scale by 255.49 and truncate.
    static float toSRGB(float x) { return sqrtf(x); }
    void fromLinear(float r, float g, float b) {
        R = static_cast<uint8_t>(255.49f * saturate(toSRGB(r)));
        G = static_cast<uint8_t>(255.49f * saturate(toSRGB(g)));
        B = static_cast<uint8_t>(255.49f * saturate(toSRGB(b)));
        A = 255;
    }
};

struct Pixel {
    uint16_t x, y;
    Color color;
};

struct Framebuffer {
    uint16_t width = 0;
    uint16_t height = 0;
    uint8_t* colorBuffer = nullptr;
    float* accumulationBuffer = nullptr;

    void create(int w, int h) {
        width = static_cast<uint16_t>(w);
        height = static_cast<uint16_t>(h);
        const size_t n = static_cast<size_t>(width) * height;
        colorBuffer = new uint8_t[n * sizeof(Color)];
        accumulationBuffer = new float[n * 3];
        memset(colorBuffer, 0, n * sizeof(Color));
        memset(accumulationBuffer, 0, n * 3 * sizeof(float));
    }

    void destroy() {
        delete[] colorBuffer;
        delete[] accumulationBuffer;
        colorBuffer = nullptr;
        accumulationBuffer = nullptr;
    }

    // Takes a width*height RGBA32F image (what pt_readback_rgba32f returns: the running MEAN, row 0 = bottom)
    // and stores it as the reference's CPU path would hold it after `samples` passes: accumulationBuffer = running
    // SUM (mean * samples), colorBuffer = display colour of the mean written in B,G,R order
    // (PathTracer.cpp:1356-1364 passes color.z, color.y, color.x).
    void storeMean(const float* rgba, int samples) {
        const size_t n = static_cast<size_t>(width) * height;
        Color* out = reinterpret_cast<Color*>(colorBuffer);
        const float s = static_cast<float>(samples);
        for (size_t i = 0; i < n; ++i) {
            const float r = rgba[4 * i], g = rgba[4 * i + 1], b = rgba[4 * i + 2];
            accumulationBuffer[3 * i] = r * s;
            accumulationBuffer[3 * i + 1] = g * s;
            accumulationBuffer[3 * i + 2] = b * s;
            out[i].fromLinear(b, g, r);
        }
    }

    // For PT_PROGRAM_CPU_TRACER, whose target holds running SUMS like the reference's CPU path: stores a width*height RGBA32F
    // image of sums after `samples` frames exactly where that path keeps things (PathTracer.cpp:1319-1368). The frame loop
    // walks 16x16 tiles (th, tw, then rows j, then pixels i) and advances ONE pointer through accumulationBuffer, so
    // accumulationBuffer is in tile-traversal order: entry k belongs to the k-th pixel visited. colorBuffer is in pixel order
    // (row j + 16 th, column i + 16 tw) and receives fromLinear(z, y, x) of sum / samples. Pixels outside the whole tiles are
    // never visited by the reference (width / 16 and height / 16 round down) and stay as they are.
    void storeSumsTileOrder(const float* rgba_sums, int samples) {
        const int tile = 16, tiles_w = width / tile, tiles_h = height / tile;
        Color* out = reinterpret_cast<Color*>(colorBuffer);
        const float inv = 1.f / static_cast<float>(samples);
        float* accum = accumulationBuffer;
        for (int th = 0; th < tiles_h; ++th)
            for (int tw = 0; tw < tiles_w; ++tw)
                for (int j = 0; j < tile; ++j)
                    for (int i = 0; i < tile; ++i) {
                        const size_t pix = static_cast<size_t>(width) * (tile * th + j) + (tile * tw + i);
                        const float x = rgba_sums[4 * pix], y = rgba_sums[4 * pix + 1], z = rgba_sums[4 * pix + 2];
                        accum[0] = x; accum[1] = y; accum[2] = z;
                        accum += 3;
                        out[pix].fromLinear(z * inv, y * inv, x * inv);
                    }
    }
};
