// Screen::writeBitmapToFile (src/screen.cpp:38-49): clamp to [0,1], * 255 with truncation to u8, BMP on disk.
#include "screen.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <thread>

void Screen::writeBitmapToFile(const std::filesystem::path& filePath) const {
    const int rowBytes = (m_w * 3 + 3) & ~3;
    const uint32_t dataSize = (uint32_t)rowBytes * (uint32_t)m_h, fileSize = 54 + dataSize;
    uint8_t hdr[54] = {'B', 'M'};
    auto put32 = [&](int off, uint32_t v) {
        for (int i = 0; i < 4; i++) hdr[off + i] = (uint8_t)(v >> (8 * i));
    };
    put32(2, fileSize);
    put32(10, 54);
    put32(14, 40);
    put32(18, (uint32_t)m_w);
    put32(22, (uint32_t)m_h);
    hdr[26] = 1;
    hdr[28] = 24;
    put32(34, dataSize);
    std::ofstream f(filePath, std::ios::binary);
    f.write(reinterpret_cast<const char*>(hdr), 54);
    std::vector<uint8_t> row(rowBytes, 0);
    for (int y = m_h - 1; y >= 0; y--) {  // m_data row 0 is the TOP of the image (setPixel flips), BMP stores bottom-up
        for (int x = 0; x < m_w; x++) {
            const cgrt::vec3& c = m_data[(size_t)y * m_w + x];
            const float ch[3] = {c.z, c.y, c.x};  // BGR
            for (int k = 0; k < 3; k++) row[3 * x + k] = (uint8_t)(std::min(std::max(ch[k], 0.0f), 1.0f) * 255.0f);
        }
        f.write(reinterpret_cast<const char*>(row.data()), rowBytes);
    }
}

void Screen::setFrame(const float* rgb) {
    static_assert(sizeof(cgrt::vec3) == 12, "vec3 is three packed floats");
    auto rows = [&](int y0, int y1) {
        for (int y = y0; y < y1; y++) std::memcpy(static_cast<void*>(&m_data[(size_t)(m_h - 1 - y) * m_w]), rgb + 3 * (size_t)y * m_w, (size_t)m_w * 12);
    };
    const size_t bytes = (size_t)m_w * m_h * 12;
    const int nt = bytes < (8u << 20) ? 1 : 4;
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(rows, m_h * t / nt, m_h * (t + 1) / nt);
    rows(0, m_h / nt);
    for (std::thread& th : pool) th.join();
}
