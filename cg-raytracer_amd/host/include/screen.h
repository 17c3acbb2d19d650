// Mirror of src/screen.h without OpenGL: float framebuffer, y-flip on setPixel, clamp -> u8 BMP (screen.cpp:30-49).
#pragma once
#include <filesystem>
#include <vector>

#include "cgrt_vec.h"

class Screen {
public:
    Screen(int width, int height) : m_w(width), m_h(height), m_data((size_t)width * height) {}
    void clear(const cgrt::vec3& color) { std::fill(m_data.begin(), m_data.end(), color); }
    void setPixel(int x, int y, const cgrt::vec3& color) { m_data[(size_t)(m_h - 1 - y) * m_w + x] = color; }  // screen.cpp:30-36
    // setPixel(x, y, rgb[y*W + x]) for the whole frame (row y of `rgb` lands in row H-1-y: the same flip), rows copied in bulk
    // on a few threads: what the device drivers use to fill the Screen from the library's pinned frame
    void setFrame(const float* rgb);
    void writeBitmapToFile(const std::filesystem::path& filePath) const;                                      // screen.cpp:38-49
    const std::vector<cgrt::vec3>& pixels() const { return m_data; }
    int width() const { return m_w; }
    int height() const { return m_h; }

private:
    int m_w, m_h;
    std::vector<cgrt::vec3> m_data;
};
