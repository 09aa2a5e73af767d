// CPU-side image store with the reference's Framebuffer layout (Framebuffer.h:12-58 in the reference):
//   colorBuffer         width*height RGBA8 display pixels
//   accumulationBuffer  width*height*3 floats, zero-initialised linear radiance
// The HIP kernels accumulate in device memory; this struct is where a host reads the result back to, in the
// layout existing code expects.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

inline float saturate(float f) { return f < 0.f ? 0.f : (f > 1.f ? 1.f : f); }

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
};
