// ss_tables.hpp — host-side construction of the constant tables the kernels read (double precision -> f32).
// Layout of the returned buffer (floats): twM[2*1024] | twItem[2*2048] | tw512[2*256] | win[512] | twG[2*512] | twP2[2*512].
#pragma once
#include <cmath>
#include <vector>

namespace ssk_host {

constexpr int kTwMOff = 0;
constexpr int kTwItemOff = 2 * 1024;
constexpr int kTw512Off = kTwItemOff + 2 * 2048;
constexpr int kWinOff = kTw512Off + 2 * 256;
constexpr int kTwGOff = kWinOff + 512;
constexpr int kTwP2Off = kTwGOff + 2 * 512;
constexpr int kTableFloats = kTwP2Off + 2 * 512;

inline std::vector<float> build_tables() {
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<float> host(kTableFloats);
    float* twM = host.data() + kTwMOff;
    float* twItem = host.data() + kTwItemOff;
    float* tw512 = host.data() + kTw512Off;
    float* win = host.data() + kWinOff;
    for (int t = 0; t < 1024; ++t) {                       // exp(-2 pi i t / 16384)
        const double a = -two_pi * t / 16384.0;
        twM[2 * t] = static_cast<float>(std::cos(a));
        twM[2 * t + 1] = static_cast<float>(std::sin(a));
    }
    for (int q = 0; q < 2048; ++q) {                       // exp(-2 pi i gA(q) / 32768), mirrors ssk::item_gA
        const int c = q & 15, lo = q >> 4;
        const int g = lo != 0 ? lo + 256 * c : (c < 8 ? 256 * c : 128 + 256 * (c - 8));
        const double a = -two_pi * g / 32768.0;
        twItem[2 * q] = static_cast<float>(std::cos(a));
        twItem[2 * q + 1] = static_cast<float>(std::sin(a));
    }
    for (int k = 0; k < 256; ++k) {                        // exp(-2 pi i k / 512)
        const double a = -two_pi * k / 512.0;
        tw512[2 * k] = static_cast<float>(std::cos(a));
        tw512[2 * k + 1] = static_cast<float>(std::sin(a));
    }
    // scipy.signal.get_window('hann', 400, fftbins=True) centre-padded to n_fft = 512 (librosa.stft)
    for (int n = 0; n < 512; ++n) {
        const int j = n - 56;
        win[n] = (j >= 0 && j < 400) ? static_cast<float>(0.5 - 0.5 * std::cos(two_pi * j / 400.0)) : 0.f;
    }
    float* twG = host.data() + kTwGOff;
    float* twP2 = host.data() + kTwP2Off;
    for (int q = 0; q < 512; ++q) {                        // exp(-2 pi i q / 32768): Hermitian stage of the 512-thread core
        const double a = -two_pi * q / 32768.0;
        twG[2 * q] = static_cast<float>(std::cos(a));
        twG[2 * q + 1] = static_cast<float>(std::sin(a));
    }
    for (int c = 0; c < 16; ++c)                           // exp(-2 pi i c k2 / 512): its pass-2 twiddles
        for (int k2 = 0; k2 < 32; ++k2) {
            const double a = -two_pi * (c * k2) / 512.0;
            twP2[2 * (c * 32 + k2)] = static_cast<float>(std::cos(a));
            twP2[2 * (c * 32 + k2) + 1] = static_cast<float>(std::sin(a));
        }
    return host;
}

}  // namespace ssk_host
