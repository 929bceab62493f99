// h264_ingest.cpp -- the C-ABI video ingest (include/lkingest.h): ISO-BMFF sample tables + an H.264 Constrained-Baseline decoder for
// hosts without cv::VideoCapture.  A second, scalar implementation of optical-flow_amd/h264_ingest.py (the clause numbers of ITU-T
// H.264 (03/2005) are given where each piece starts); the two must agree byte for byte (tests/test_h264_ingest.py).  Host code only.
#include "../../include/lkingest.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };
struct BadStream : std::runtime_error { using std::runtime_error::runtime_error; };

// ---------------------------------------------------------------------------------------------- bit reader (7.2, 9.1)
struct Bits {
    std::vector<uint8_t> d;      // RBSP (emulation-prevention bytes removed)
    size_t pos = 0, n = 0, end = 0;
    explicit Bits(const uint8_t *nal, size_t len)
    {
        int z = 0;
        for (size_t i = 0; i < len; ++i) {
            if (z >= 2 && nal[i] == 3) { z = 0; continue; }
            d.push_back(nal[i]);
            z = nal[i] == 0 ? z + 1 : 0;
        }
        n = d.size() * 8;
        end = 0;
        for (size_t i = d.size(); i-- > 0;)
            if (d[i]) { int b = 0; while (!((d[i] >> b) & 1)) ++b; end = i * 8 + (7 - b); break; }   // the rbsp_stop_one_bit
        d.resize(d.size() + 8, 0);
    }
    unsigned bit() { if (pos >= n) throw BadStream("read past the end of the NAL unit"); unsigned r = (d[pos >> 3] >> (7 - (pos & 7))) & 1; ++pos; return r; }
    unsigned u(int k) { unsigned r = 0; for (int i = 0; i < k; ++i) r = (r << 1) | bit(); return r; }
    unsigned peek(int k) const
    {
        unsigned r = 0;
        for (int i = 0; i < k; ++i) { size_t p = pos + i; r = (r << 1) | (p < n ? (d[p >> 3] >> (7 - (p & 7))) & 1 : 0); }
        return r;
    }
    unsigned ue() { int z = 0; while (bit() == 0) if (++z > 31) throw BadStream("bad Exp-Golomb code"); return (1u << z) - 1 + u(z); }
    int se() { unsigned k = ue(); return (k & 1) ? (int)((k + 1) >> 1) : -(int)(k >> 1); }
    int te(int cmax) { return cmax > 1 ? (int)ue() : 1 - (int)bit(); }
    bool more() const { return pos < end; }
    void align() { pos = (pos + 7) & ~(size_t)7; }
};

// ---------------------------------------------------------------------------------------------- CAVLC tables (9.2)
const uint8_t CT_LEN[3][68] = {
    {1, 0, 0, 0, 6, 2, 0, 0, 8, 6, 3, 0, 9, 8, 7, 5, 10, 9, 8, 6, 11, 10, 9, 7, 13, 11, 10, 8, 13, 13, 11, 9, 13, 13, 13, 10,
     14, 14, 13, 11, 14, 14, 14, 13, 15, 15, 14, 14, 15, 15, 15, 14, 16, 15, 15, 15, 16, 16, 16, 15, 16, 16, 16, 16, 16, 16, 16, 16},
    {2, 0, 0, 0, 6, 2, 0, 0, 6, 5, 3, 0, 7, 6, 6, 4, 8, 6, 6, 4, 8, 7, 7, 5, 9, 8, 8, 6, 11, 9, 9, 6, 11, 11, 11, 7,
     12, 11, 11, 9, 12, 12, 12, 11, 12, 12, 12, 11, 13, 13, 13, 12, 13, 13, 13, 13, 13, 14, 13, 13, 14, 14, 14, 13, 14, 14, 14, 14},
    {4, 0, 0, 0, 6, 4, 0, 0, 6, 5, 4, 0, 6, 5, 5, 4, 7, 5, 5, 4, 7, 5, 5, 4, 7, 6, 6, 4, 7, 6, 6, 4, 8, 7, 7, 5,
     8, 8, 7, 6, 9, 8, 8, 7, 9, 9, 8, 8, 9, 9, 9, 8, 10, 9, 9, 9, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10}};
const uint8_t CT_BITS[3][68] = {
    {1, 0, 0, 0, 5, 1, 0, 0, 7, 4, 1, 0, 7, 6, 5, 3, 7, 6, 5, 3, 7, 6, 5, 4, 15, 6, 5, 4, 11, 14, 5, 4, 8, 10, 13, 4,
     15, 14, 9, 4, 11, 10, 13, 12, 15, 14, 9, 12, 11, 10, 13, 8, 15, 1, 9, 12, 11, 14, 13, 8, 7, 10, 9, 12, 4, 6, 5, 8},
    {3, 0, 0, 0, 11, 2, 0, 0, 7, 7, 3, 0, 7, 10, 9, 5, 7, 6, 5, 4, 4, 6, 5, 6, 7, 6, 5, 8, 15, 6, 5, 4, 11, 14, 13, 4,
     15, 10, 9, 4, 11, 14, 13, 12, 8, 10, 9, 8, 15, 14, 13, 12, 11, 10, 9, 12, 7, 11, 6, 8, 9, 8, 10, 1, 7, 6, 5, 4},
    {15, 0, 0, 0, 15, 14, 0, 0, 11, 15, 13, 0, 8, 12, 14, 12, 15, 10, 11, 11, 11, 8, 9, 10, 9, 14, 13, 9, 8, 10, 9, 8, 15, 14, 13, 13,
     11, 14, 10, 12, 15, 10, 13, 12, 11, 14, 9, 12, 8, 10, 13, 8, 13, 7, 9, 12, 9, 12, 11, 10, 5, 8, 7, 6, 1, 4, 3, 2}};
const uint8_t CDC_LEN[20] = {2, 0, 0, 0, 6, 1, 0, 0, 6, 6, 3, 0, 6, 7, 7, 6, 6, 8, 8, 7};
const uint8_t CDC_BITS[20] = {1, 0, 0, 0, 7, 1, 0, 0, 4, 6, 1, 0, 3, 3, 2, 5, 2, 3, 2, 0};
const uint8_t TZ_LEN[15][16] = {
    {1, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 9}, {3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6, 6, 6}, {4, 3, 3, 3, 4, 4, 3, 3, 4, 5, 5, 6, 5, 6},
    {5, 3, 4, 4, 3, 3, 3, 4, 3, 4, 5, 5, 5}, {4, 4, 4, 3, 3, 3, 3, 3, 4, 5, 4, 5}, {6, 5, 3, 3, 3, 3, 3, 3, 4, 3, 6}, {6, 5, 3, 3, 3, 2, 3, 4, 3, 6},
    {6, 4, 5, 3, 2, 2, 3, 3, 6}, {6, 6, 4, 2, 2, 3, 2, 5}, {5, 5, 3, 2, 2, 2, 4}, {4, 4, 3, 3, 1, 3}, {4, 4, 2, 1, 3}, {3, 3, 1, 2}, {2, 2, 1}, {1, 1}};
const uint8_t TZ_BITS[15][16] = {
    {1, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 1}, {7, 6, 5, 4, 3, 5, 4, 3, 2, 3, 2, 3, 2, 1, 0}, {5, 7, 6, 5, 4, 3, 4, 3, 2, 3, 2, 1, 1, 0},
    {3, 7, 5, 4, 6, 5, 4, 3, 3, 2, 2, 1, 0}, {5, 4, 3, 7, 6, 5, 4, 3, 2, 1, 1, 0}, {1, 1, 7, 6, 5, 4, 3, 2, 1, 1, 0}, {1, 1, 5, 4, 3, 3, 2, 1, 1, 0},
    {1, 1, 1, 3, 3, 2, 2, 1, 0}, {1, 0, 1, 3, 2, 1, 1, 1}, {1, 0, 1, 3, 2, 1, 1}, {0, 1, 1, 2, 1, 3}, {0, 1, 1, 1, 1}, {0, 1, 1, 1}, {0, 1, 1}, {0, 1}};
const uint8_t TZ_N[15] = {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2};
const uint8_t CTZ_LEN[3][4] = {{1, 2, 3, 3}, {1, 2, 2}, {1, 1}};
const uint8_t CTZ_BITS[3][4] = {{1, 1, 1, 0}, {1, 1, 0}, {1, 0}};
const uint8_t CTZ_N[3] = {4, 3, 2};
const uint8_t RUN_LEN[7][16] = {{1, 1}, {1, 2, 2}, {2, 2, 2, 2}, {2, 2, 2, 3, 3}, {2, 2, 3, 3, 3, 3}, {2, 3, 3, 3, 3, 3, 3}, {3, 3, 3, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9, 10, 11}};
const uint8_t RUN_BITS[7][16] = {{1, 0}, {1, 1, 0}, {3, 2, 1, 0}, {3, 2, 1, 1, 0}, {3, 2, 3, 2, 1, 0}, {3, 0, 1, 3, 2, 5, 4}, {7, 6, 5, 4, 3, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1}};
const uint8_t RUN_N[7] = {2, 3, 4, 5, 6, 7, 15};

int read_vlc(Bits &b, const uint8_t *len, const uint8_t *code, int n, const char *name)
{
    for (int s = 0; s < n; ++s)
        if (len[s] && b.peek(len[s]) == code[s]) { b.pos += len[s]; return s; }
    throw BadStream(std::string("invalid code in ") + name);
}

const uint8_t CBP_INTRA[48] = {47, 31, 15, 0, 23, 27, 29, 30, 7, 11, 13, 14, 39, 43, 45, 46, 16, 3, 5, 10, 12, 19, 21, 26, 28, 35, 37, 42, 44, 1, 2, 4, 8, 17, 18, 20, 24,
                               6, 9, 22, 25, 32, 33, 34, 36, 40, 38, 41};
const uint8_t CBP_INTER[48] = {0, 16, 1, 2, 4, 8, 32, 3, 5, 10, 12, 15, 47, 7, 11, 13, 14, 6, 9, 31, 35, 37, 42, 44, 33, 34, 36, 40, 39, 43, 45, 46, 17, 18, 20, 24,
                               19, 21, 26, 28, 23, 27, 29, 30, 22, 25, 38, 41};
const uint8_t ZZ_X[16] = {0, 1, 0, 0, 1, 2, 3, 2, 1, 0, 1, 2, 3, 3, 2, 3};
const uint8_t ZZ_Y[16] = {0, 0, 1, 2, 1, 0, 0, 1, 2, 3, 3, 2, 1, 2, 3, 3};
const uint8_t BLK_X[16] = {0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3};
const uint8_t BLK_Y[16] = {0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3};
const int NORM_V[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16}, {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};
inline int level_scale(int m, int x, int y) { return 16 * (((x | y) & 1) == 0 ? NORM_V[m][0] : ((x & y & 1) ? NORM_V[m][1] : NORM_V[m][2])); }
const uint8_t QPC[52] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
                         29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};
const uint8_t ALPHA[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90,
                           101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
const uint8_t BETA[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15,
                          16, 16, 17, 17, 18, 18};
const uint8_t TC0[52][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0},
                            {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 1, 1}, {0, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 1, 1},
                            {1, 1, 1}, {1, 1, 2}, {1, 1, 2}, {1, 1, 2}, {1, 1, 2}, {1, 2, 3}, {1, 2, 3}, {2, 2, 3}, {2, 2, 4}, {2, 3, 4}, {2, 3, 4}, {3, 3, 5}, {3, 4, 6},
                            {3, 4, 6}, {4, 5, 7}, {4, 5, 8}, {4, 6, 9}, {5, 7, 10}, {6, 8, 11}, {6, 8, 13}, {7, 10, 14}, {8, 11, 16}, {9, 12, 18}, {10, 13, 20},
                            {11, 15, 23}, {13, 17, 25}};

inline int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

// 7.3.5.3.1 / 9.2: coefficients in scan order; returns TotalCoeff
int residual_block(Bits &b, int nC, int max_coeff, int *coef)
{
    int tok;
    if (nC < 0) tok = read_vlc(b, CDC_LEN, CDC_BITS, 20, "coeff_token (chroma DC)");
    else if (nC < 2) tok = read_vlc(b, CT_LEN[0], CT_BITS[0], 68, "coeff_token");
    else if (nC < 4) tok = read_vlc(b, CT_LEN[1], CT_BITS[1], 68, "coeff_token");
    else if (nC < 8) tok = read_vlc(b, CT_LEN[2], CT_BITS[2], 68, "coeff_token");
    else { unsigned v = b.u(6); tok = v == 3 ? 0 : (int)((v >> 2) + 1) * 4 + (int)(v & 3); }
    const int total = tok >> 2, t1 = tok & 3;
    for (int i = 0; i < max_coeff; ++i) coef[i] = 0;
    if (total == 0) return 0;
    if (total > max_coeff || t1 > total) throw BadStream("coeff_token out of range");
    int level[16], run[16];
    int suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
    for (int i = 0; i < total; ++i) {
        if (i < t1) { level[i] = 1 - 2 * (int)b.bit(); continue; }
        int prefix = 0;
        while (b.bit() == 0) if (++prefix > 32) throw BadStream("level_prefix too long");
        int code = std::min(15, prefix) << suffix_len;
        int size = (prefix == 14 && suffix_len == 0) ? 4 : (prefix >= 15 ? prefix - 3 : suffix_len);
        if (size) code += (int)b.u(size);
        if (prefix >= 15 && suffix_len == 0) code += 15;
        if (prefix >= 16) code += (1 << (prefix - 3)) - 4096;
        if (i == t1 && t1 < 3) code += 2;
        const int lv = (code % 2 == 0) ? (code + 2) >> 1 : (-code - 1) >> 1;
        if (lv < -32768 || lv > 32767) throw BadStream("coefficient level out of range");   // (8-bit video: 16-bit levels; keeps the scaling below in range)
        level[i] = lv;
        if (suffix_len == 0) suffix_len = 1;
        if (std::abs(lv) > (3 << (suffix_len - 1)) && suffix_len < 6) ++suffix_len;
    }
    int zeros_left = 0;
    if (total < max_coeff) {
        if (max_coeff == 4) zeros_left = read_vlc(b, CTZ_LEN[total - 1], CTZ_BITS[total - 1], CTZ_N[total - 1], "total_zeros (chroma DC)");
        else zeros_left = read_vlc(b, TZ_LEN[total - 1], TZ_BITS[total - 1], TZ_N[total - 1], "total_zeros");
        if (zeros_left + total > max_coeff) throw BadStream("total_zeros out of range");
    }
    for (int i = 0; i < total; ++i) run[i] = 0;
    for (int i = 0; i < total - 1; ++i)
        if (zeros_left > 0) {
            const int t = std::min(zeros_left, 7) - 1;
            run[i] = read_vlc(b, RUN_LEN[t], RUN_BITS[t], RUN_N[t], "run_before");
            zeros_left -= run[i];
            if (zeros_left < 0) throw BadStream("run_before out of range");
        }
    run[total - 1] = zeros_left;
    int k = -1;
    for (int i = total - 1; i >= 0; --i) { k += run[i] + 1; coef[k] = level[i]; }
    return total;
}

// ---------------------------------------------------------------------------------------------- transforms (8.5.10 - 8.5.12)
void itrans4x4(const int d[4][4], int r[4][4])
{
    int f[4][4], h[4][4];
    for (int y = 0; y < 4; ++y) {
        const int e0 = d[y][0] + d[y][2], e1 = d[y][0] - d[y][2], e2 = (d[y][1] >> 1) - d[y][3], e3 = d[y][1] + (d[y][3] >> 1);
        f[y][0] = e0 + e3; f[y][1] = e1 + e2; f[y][2] = e1 - e2; f[y][3] = e0 - e3;
    }
    for (int x = 0; x < 4; ++x) {
        const int g0 = f[0][x] + f[2][x], g1 = f[0][x] - f[2][x], g2 = (f[1][x] >> 1) - f[3][x], g3 = f[1][x] + (f[3][x] >> 1);
        h[0][x] = g0 + g3; h[1][x] = g1 + g2; h[2][x] = g1 - g2; h[3][x] = g0 - g3;
    }
    for (int y = 0; y < 4; ++y)
        for (int x = 0; x < 4; ++x) r[y][x] = (h[y][x] + 32) >> 6;
}

// scan-order coefficients (first = scan position `first`) -> scaled d[y][x]; the DC entry stays 0 when it comes from a DC transform
void scale4x4(const int *coef, int first, int qp, bool skip_dc, int d[4][4])
{
    for (int y = 0; y < 4; ++y) for (int x = 0; x < 4; ++x) d[y][x] = 0;
    for (int k = first; k < 16; ++k) {
        const int c = coef[k - first];
        if (!c) continue;
        const int x = ZZ_X[k], y = ZZ_Y[k], ls = level_scale(qp % 6, x, y);
        d[y][x] = qp >= 24 ? (c * ls) << (qp / 6 - 4) : (c * ls + (1 << (3 - qp / 6))) >> (4 - qp / 6);
    }
    if (skip_dc) d[0][0] = 0;
}

void luma_dc(const int *coef, int qp, int dc[4][4])
{
    static const int H4[4][4] = {{1, 1, 1, 1}, {1, 1, -1, -1}, {1, -1, -1, 1}, {1, -1, 1, -1}};
    int c[4][4] = {}, t[4][4], f[4][4];
    for (int k = 0; k < 16; ++k) c[ZZ_Y[k]][ZZ_X[k]] = coef[k];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { int s = 0; for (int k = 0; k < 4; ++k) s += H4[i][k] * c[k][j]; t[i][j] = s; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { int s = 0; for (int k = 0; k < 4; ++k) s += t[i][k] * H4[k][j]; f[i][j] = s; }
    const int64_t ls = level_scale(qp % 6, 0, 0);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const int64_t v = qp >= 36 ? (f[i][j] * ls) << (qp / 6 - 6) : (f[i][j] * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
            dc[i][j] = (int)std::clamp<int64_t>(v, -(1 << 26), 1 << 26);   // (no conforming stream comes near: keeps a damaged one inside int)
        }
}

void chroma_dc(const int *coef, int qpc, int dc[2][2])
{
    const int c00 = coef[0], c01 = coef[1], c10 = coef[2], c11 = coef[3];
    const int f[2][2] = {{c00 + c01 + c10 + c11, c00 - c01 + c10 - c11}, {c00 + c01 - c10 - c11, c00 - c01 - c10 + c11}};
    const int64_t ls = level_scale(qpc % 6, 0, 0);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) dc[i][j] = (int)std::clamp<int64_t>(((f[i][j] * ls) << (qpc / 6)) >> 5, -(1 << 26), 1 << 26);
}

// ---------------------------------------------------------------------------------------------- planes and pictures
struct Plane {
    int w = 0, h = 0;
    std::vector<int16_t> p;
    void init(int w_, int h_) { w = w_; h = h_; p.assign((size_t)w * h, 0); }
    int16_t &at(int x, int y) { return p[(size_t)y * w + x]; }
    int get(int x, int y) const { return p[(size_t)clip3(0, h - 1, y) * w + clip3(0, w - 1, x)]; }   // edge-clamped (8.4.2.2)
};

struct Picture {
    Plane Y, C[2];
    int frame_num = 0, uid = -1;
};

// 8.4.2.2.1: quarter-sample luma interpolation of a w x h block
void mc_luma(const Plane &ref, int x, int y, int w, int h, int mvx, int mvy, int *out /* [h][w] */)
{
    const int dx = mvx & 3, dy = mvy & 3, x0 = x + (mvx >> 2), y0 = y + (mvy >> 2);
    auto G = [&](int xx, int yy) { return ref.get(x0 + xx, y0 + yy); };
    auto tap = [](int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; };
    auto b1 = [&](int xx, int yy) { return tap(G(xx - 2, yy), G(xx - 1, yy), G(xx, yy), G(xx + 1, yy), G(xx + 2, yy), G(xx + 3, yy)); };   // half sample right of (xx, yy)
    auto h1 = [&](int xx, int yy) { return tap(G(xx, yy - 2), G(xx, yy - 1), G(xx, yy), G(xx, yy + 1), G(xx, yy + 2), G(xx, yy + 3)); };   // half sample below
    auto bb = [&](int xx, int yy) { return clip8((b1(xx, yy) + 16) >> 5); };
    auto hh = [&](int xx, int yy) { return clip8((h1(xx, yy) + 16) >> 5); };
    auto jj = [&](int xx, int yy) { return clip8((tap(b1(xx, yy - 2), b1(xx, yy - 1), b1(xx, yy), b1(xx, yy + 1), b1(xx, yy + 2), b1(xx, yy + 3)) + 512) >> 10); };
    for (int yy = 0; yy < h; ++yy)
        for (int xx = 0; xx < w; ++xx) {
            int v;
            if (dx == 0 && dy == 0) v = G(xx, yy);
            else if (dy == 0) v = dx == 2 ? bb(xx, yy) : ((dx == 1 ? G(xx, yy) : G(xx + 1, yy)) + bb(xx, yy) + 1) >> 1;
            else if (dx == 0) v = dy == 2 ? hh(xx, yy) : ((dy == 1 ? G(xx, yy) : G(xx, yy + 1)) + hh(xx, yy) + 1) >> 1;
            else if (dx == 2 && dy == 2) v = jj(xx, yy);
            else if (dx == 2) v = (jj(xx, yy) + (dy == 1 ? bb(xx, yy) : bb(xx, yy + 1)) + 1) >> 1;
            else if (dy == 2) v = (jj(xx, yy) + (dx == 1 ? hh(xx, yy) : hh(xx + 1, yy)) + 1) >> 1;
            else v = ((dy == 1 ? bb(xx, yy) : bb(xx, yy + 1)) + (dx == 1 ? hh(xx, yy) : hh(xx + 1, yy)) + 1) >> 1;
            out[yy * w + xx] = v;
        }
}

void mc_chroma(const Plane &ref, int x, int y, int w, int h, int mvx, int mvy, int *out)
{
    const int dx = mvx & 7, dy = mvy & 7, x0 = x + (mvx >> 3), y0 = y + (mvy >> 3);
    for (int yy = 0; yy < h; ++yy)
        for (int xx = 0; xx < w; ++xx) {
            const int A = ref.get(x0 + xx, y0 + yy), B = ref.get(x0 + xx + 1, y0 + yy), C = ref.get(x0 + xx, y0 + yy + 1), D = ref.get(x0 + xx + 1, y0 + yy + 1);
            out[yy * w + xx] = ((8 - dx) * (8 - dy) * A + dx * (8 - dy) * B + (8 - dx) * dy * C + dx * dy * D + 32) >> 6;
        }
}

// ---------------------------------------------------------------------------------------------- intra prediction (8.3)
inline int f3(int a, int b, int c) { return (a + 2 * b + c + 2) >> 2; }

// top: 8 samples or nullptr, left: 4 or nullptr, tl valid when both are
void pred_intra4x4(int mode, const int *top, const int *left, int tl, int p[4][4])
{
    if (mode == 2) {
        int v;
        if (top && left) v = (top[0] + top[1] + top[2] + top[3] + left[0] + left[1] + left[2] + left[3] + 4) >> 3;
        else if (left) v = (left[0] + left[1] + left[2] + left[3] + 2) >> 2;
        else if (top) v = (top[0] + top[1] + top[2] + top[3] + 2) >> 2;
        else v = 128;
        for (int y = 0; y < 4; ++y) for (int x = 0; x < 4; ++x) p[y][x] = v;
        return;
    }
    if (((mode == 0 || mode == 3 || mode == 7) && !top) || ((mode == 1 || mode == 8) && !left) || ((mode >= 4 && mode <= 6) && !(top && left)))
        throw BadStream("intra 4x4 mode needs unavailable neighbours");
    int e[9] = {0};
    if (top && left) { e[0] = left[3]; e[1] = left[2]; e[2] = left[1]; e[3] = left[0]; e[4] = tl; e[5] = top[0]; e[6] = top[1]; e[7] = top[2]; e[8] = top[3]; }
    for (int y = 0; y < 4; ++y)
        for (int x = 0; x < 4; ++x) {
            int v = 0;
            switch (mode) {
            case 0: v = top[x]; break;
            case 1: v = left[y]; break;
            case 3: v = (x == 3 && y == 3) ? (top[6] + 3 * top[7] + 2) >> 2 : f3(top[x + y], top[x + y + 1], top[x + y + 2]); break;
            case 7: { const int k = x + (y >> 1); v = (y % 2 == 0) ? (top[k] + top[k + 1] + 1) >> 1 : f3(top[k], top[k + 1], top[k + 2]); break; }
            case 8: {
                const int z = x + 2 * y, k = y + (x >> 1);
                if (z > 5) v = left[3];
                else if (z == 5) v = (left[2] + 3 * left[3] + 2) >> 2;
                else if (z % 2 == 0) v = (left[k] + left[k + 1] + 1) >> 1;
                else v = f3(left[k], left[k + 1], left[k + 2]);
                break;
            }
            case 4: { const int i = 4 + x - y; v = f3(e[i - 1], e[i], e[i + 1]); break; }
            case 5: {
                const int z = 2 * x - y, k = x - (y >> 1);
                if (z >= 0 && z % 2 == 0) v = (e[5 + k - 1] + e[5 + k] + 1) >> 1;
                else if (z >= 0) v = f3(e[5 + k - 2], e[5 + k - 1], e[5 + k]);
                else if (z == -1) v = f3(e[3], e[4], e[5]);
                else v = f3(e[3 - (y - 1)], e[3 - (y - 2)], e[3 - (y - 3)]);
                break;
            }
            case 6: {
                const int z = 2 * y - x, k = y - (x >> 1);
                if (z >= 0 && z % 2 == 0) v = (e[3 - (k - 1)] + e[3 - k] + 1) >> 1;
                else if (z >= 0) v = f3(e[3 - (k - 2)], e[3 - (k - 1)], e[3 - k]);
                else if (z == -1) v = f3(e[3], e[4], e[5]);
                else v = f3(e[5 + x - 1], e[5 + x - 2], e[5 + x - 3]);
                break;
            }
            default: throw BadStream("intra 4x4 mode");
            }
            p[y][x] = v;
        }
}

// Intra16x16 / chroma plane prediction: n = 16 or 8
void pred_plane(const int *top, const int *left, int tl, int n, int *out /* [n][n] */)
{
    const int half = n / 2;
    auto T = [&](int i) { return i < 0 ? tl : top[i]; };
    auto L = [&](int i) { return i < 0 ? tl : left[i]; };
    int hs = 0, vs = 0;
    for (int k = 0; k < half; ++k) { hs += (k + 1) * (T(half + k) - T(half - 2 - k)); vs += (k + 1) * (L(half + k) - L(half - 2 - k)); }
    const int a = 16 * (left[n - 1] + top[n - 1]);
    const int b = n == 16 ? (5 * hs + 32) >> 6 : (34 * hs + 32) >> 6, c = n == 16 ? (5 * vs + 32) >> 6 : (34 * vs + 32) >> 6;
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) out[y * n + x] = clip8((a + b * (x - (half - 1)) + c * (y - (half - 1)) + 16) >> 5);
}

struct SPS { int log2_max_frame_num = 4, poc_type = 0, log2_max_poc_lsb = 4, delta_always_zero = 0, num_ref_frames = 1, mb_w = 0, mb_h = 0, crop[4] = {0, 0, 0, 0}; };
struct PPS { int id = 0, pic_order_present = 0, num_ref_idx = 1, init_qp = 26, chroma_qp_offset = 0, deblock_control = 0, redundant_pic_cnt = 0; };

SPS parse_sps(const uint8_t *nal, size_t len)
{
    Bits b(nal + 1, len - 1);
    SPS s;
    const unsigned profile = b.u(8);
    b.u(8); b.u(8); b.ue();
    if (profile == 100 || profile == 110 || profile == 122 || profile == 244 || profile == 44 || profile == 83 || profile == 86 || profile == 118 || profile == 128)
        throw Unsupported("High profiles");
    s.log2_max_frame_num = (int)b.ue() + 4;
    s.poc_type = (int)b.ue();
    if (s.poc_type == 0) s.log2_max_poc_lsb = (int)b.ue() + 4;
    else if (s.poc_type == 1) { s.delta_always_zero = (int)b.bit(); b.se(); b.se(); for (unsigned k = b.ue(); k > 0; --k) b.se(); }
    s.num_ref_frames = (int)b.ue();
    b.bit();
    s.mb_w = (int)b.ue() + 1;
    s.mb_h = (int)b.ue() + 1;
    if (!b.bit()) throw Unsupported("interlace (frame_mbs_only_flag = 0)");
    b.bit();
    if (b.bit()) for (int k = 0; k < 4; ++k) s.crop[k] = (int)b.ue();
    return s;
}

PPS parse_pps(const uint8_t *nal, size_t len)
{
    Bits b(nal + 1, len - 1);
    PPS p;
    p.id = (int)b.ue(); b.ue();
    if (b.bit()) throw Unsupported("CABAC");
    p.pic_order_present = (int)b.bit();
    if (b.ue() != 0) throw Unsupported("slice groups (FMO)");
    p.num_ref_idx = (int)b.ue() + 1;
    b.ue();
    if (b.bit() || b.u(2)) throw Unsupported("weighted prediction");
    p.init_qp = 26 + b.se();
    b.se();
    p.chroma_qp_offset = b.se();
    p.deblock_control = (int)b.bit();
    b.bit();
    p.redundant_pic_cnt = (int)b.bit();
    return p;
}

struct SliceHdr { int type = 0, frame_num = 0, num_ref = 1, qp = 26, deblock = 0, alpha_off = 0, beta_off = 0; bool idr = false; std::vector<std::pair<int, int>> reorder, mmco; };

struct MV { int x = 0, y = 0; bool operator==(const MV &o) const { return x == o.x && y == o.y; } };

struct Decoder {
    SPS sps; PPS pps;
    std::vector<std::shared_ptr<Picture>> refs;   // short-term reference pictures, most recent first
    int npic = 0;

    SliceHdr slice_header(Bits &b, int nal_type, int nal_ref_idc)
    {
        SliceHdr h;
        const unsigned first_mb = b.ue();
        h.type = (int)(b.ue() % 5);
        if (h.type != 0 && h.type != 2) throw Unsupported("slice type");
        if ((int)b.ue() != pps.id) throw Unsupported("several picture parameter sets");
        h.frame_num = (int)b.u(sps.log2_max_frame_num);
        if (nal_type == 5) b.ue();
        if (sps.poc_type == 0) { b.u(sps.log2_max_poc_lsb); if (pps.pic_order_present) b.se(); }
        else if (sps.poc_type == 1 && !sps.delta_always_zero) { b.se(); if (pps.pic_order_present) b.se(); }
        if (pps.redundant_pic_cnt) b.ue();
        h.num_ref = pps.num_ref_idx;
        if (h.type == 0) {
            if (b.bit()) h.num_ref = (int)b.ue() + 1;
            if (b.bit())
                for (;;) { const int idc = (int)b.ue(); if (idc == 3) break; h.reorder.push_back({idc, (int)b.ue()}); }
        }
        h.idr = nal_type == 5;
        if (nal_ref_idc) {
            if (nal_type == 5) { b.bit(); if (b.bit()) throw Unsupported("long-term reference"); }
            else if (b.bit())
                for (;;) { const int op = (int)b.ue(); if (op == 0) break; if (op != 1) throw Unsupported("memory_management_control_operation"); h.mmco.push_back({op, (int)b.ue()}); }
        }
        h.qp = pps.init_qp + b.se();
        if (h.qp < 0 || h.qp > 51) throw BadStream("slice QP out of range");
        if (pps.deblock_control) { h.deblock = (int)b.ue(); if (h.deblock != 1) { h.alpha_off = 2 * b.se(); h.beta_off = 2 * b.se(); } }
        if (first_mb != 0) throw Unsupported("several slices per picture");
        return h;
    }

    std::vector<std::shared_ptr<Picture>> ref_list(const SliceHdr &h)
    {
        const int max_fn = 1 << sps.log2_max_frame_num;
        auto picnum = [&](const Picture &p) { return p.frame_num <= h.frame_num ? p.frame_num : p.frame_num - max_fn; };
        std::vector<std::shared_ptr<Picture>> lst = refs;
        std::stable_sort(lst.begin(), lst.end(), [&](const auto &a, const auto &b) { return picnum(*a) > picnum(*b); });
        int pred = h.frame_num;
        for (size_t k = 0; k < h.reorder.size(); ++k) {
            const int idc = h.reorder[k].first, v = h.reorder[k].second;
            if (idc > 1) throw Unsupported("long-term reordering");
            pred = idc == 0 ? ((pred - (v + 1)) % max_fn + max_fn) % max_fn : (pred + (v + 1)) % max_fn;
            const int want = pred <= h.frame_num ? pred : pred - max_fn;
            std::shared_ptr<Picture> pic;
            for (auto &p : lst) if (picnum(*p) == want) { pic = p; break; }
            if (!pic) throw BadStream("reordering names a missing picture");
            std::vector<std::shared_ptr<Picture>> nl(lst.begin(), lst.begin() + std::min(k, lst.size()));
            nl.push_back(pic);
            for (size_t i = std::min(k, lst.size()); i < lst.size(); ++i) if (lst[i] != pic) nl.push_back(lst[i]);
            lst = nl;
        }
        if ((int)lst.size() > h.num_ref) lst.resize(h.num_ref);
        if (h.type == 0 && (int)lst.size() < h.num_ref) throw BadStream("missing reference picture");
        return lst;
    }

    // per-picture macroblock state
    int mb_w = 0, mb_h = 0, W4 = 0, H4 = 0;
    std::vector<int> nnzY, nnzC[2], ipm, refi, refu, mb_qp;
    std::vector<MV> mvs;
    std::vector<char> mb_intra;

    std::shared_ptr<Picture> decode_nal(const uint8_t *nal, size_t len);
    void decode_slice(Bits &b, const SliceHdr &h, Picture &pic, const std::vector<std::shared_ptr<Picture>> &reflist);
    void deblock(Picture &pic, const SliceHdr &h);
};

// ---------------------------------------------------------------------------------------------- slice data (7.3.4, 7.3.5, 8.3, 8.4)
inline int next_qp(int qp, int delta)      // 7.4.5: mb_qp_delta is in -26 .. 25
{
    if (delta < -26 || delta > 25) throw BadStream("mb_qp_delta out of range");
    return (qp + delta + 52) % 52;
}

void Decoder::decode_slice(Bits &b, const SliceHdr &h, Picture &pic, const std::vector<std::shared_ptr<Picture>> &reflist)
{
    const int nmb = mb_w * mb_h, C4w = W4 / 2;
    Plane &Y = pic.Y;
    int qp = h.qp;
    const bool is_p = h.type == 0;
    const int cqo = pps.chroma_qp_offset;
    auto qpc_of = [&](int q) { return (int)QPC[clip3(0, 51, q + cqo)]; };
    auto nc_of = [&](const std::vector<int> &nnz, int stride, int bx, int by) {
        const bool la = bx > 0, ta = by > 0;
        const int a = la ? nnz[(size_t)by * stride + bx - 1] : 0, t = ta ? nnz[(size_t)(by - 1) * stride + bx] : 0;
        if (la && ta) return (a + t + 1) >> 1;
        return la ? a : (ta ? t : 0);
    };
    auto avail4 = [&](int bx, int by, int cur_mb) {
        if (bx < 0 || by < 0 || bx >= W4 || by >= H4) return false;
        const int m = (by >> 2) * mb_w + (bx >> 2);
        if (m == cur_mb) return refi[(size_t)by * W4 + bx] != -2;
        return m < cur_mb;
    };
    struct Nb { bool ok; int ref; MV mv; };
    auto mv_nb = [&](int bx, int by, int cur_mb) {
        Nb n{false, -1, MV{}};
        if (!avail4(bx, by, cur_mb)) return n;
        n.ok = true;
        const int r = refi[(size_t)by * W4 + bx];
        if (r >= 0) { n.ref = r; n.mv = mvs[(size_t)by * W4 + bx]; }
        return n;
    };
    auto med = [](int a, int b_, int c) { return std::max(std::min(a, b_), std::min(std::max(a, b_), c)); };
    // shape: 0 = median only, 1 = 16x8, 2 = 8x16
    auto pred_mv = [&](int bx, int by, int w4, int ref, int cur_mb, int shape, int part) {
        const Nb A = mv_nb(bx - 1, by, cur_mb), B = mv_nb(bx, by - 1, cur_mb);
        Nb Cn = mv_nb(bx + w4, by - 1, cur_mb);
        if (!Cn.ok) Cn = mv_nb(bx - 1, by - 1, cur_mb);
        if (shape == 1) {
            if (part == 0 && B.ok && B.ref == ref) return B.mv;
            if (part == 1 && A.ok && A.ref == ref) return A.mv;
        } else if (shape == 2) {
            if (part == 0 && A.ok && A.ref == ref) return A.mv;
            if (part == 1 && Cn.ok && Cn.ref == ref) return Cn.mv;
        }
        if (!B.ok && !Cn.ok && A.ok) return A.mv;
        const int same = (A.ref == ref) + (B.ref == ref) + (Cn.ref == ref);      // (an unavailable neighbour has ref -1 and mv 0)
        if (same == 1) return A.ref == ref ? A.mv : (B.ref == ref ? B.mv : Cn.mv);
        return MV{med(A.mv.x, B.mv.x, Cn.mv.x), med(A.mv.y, B.mv.y, Cn.mv.y)};
    };
    auto set_motion = [&](int bx, int by, int w4, int h4, int ref, MV mv) {
        for (int yy = by; yy < by + h4; ++yy)
            for (int xx = bx; xx < bx + w4; ++xx) { mvs[(size_t)yy * W4 + xx] = mv; refi[(size_t)yy * W4 + xx] = ref; refu[(size_t)yy * W4 + xx] = reflist[ref]->uid; }
    };
    int tmp[256];
    auto inter_pred = [&](int bx, int by, int w4, int h4, int ref, MV mv) {
        const Picture &rp = *reflist[ref];
        const int x = 4 * bx, y = 4 * by, w = 4 * w4, hh = 4 * h4;
        mc_luma(rp.Y, x, y, w, hh, mv.x, mv.y, tmp);
        for (int yy = 0; yy < hh; ++yy) for (int xx = 0; xx < w; ++xx) Y.at(x + xx, y + yy) = (int16_t)tmp[yy * w + xx];
        for (int ci = 0; ci < 2; ++ci) {
            mc_chroma(rp.C[ci], x / 2, y / 2, w / 2, hh / 2, mv.x, mv.y, tmp);
            for (int yy = 0; yy < hh / 2; ++yy) for (int xx = 0; xx < w / 2; ++xx) pic.C[ci].at(x / 2 + xx, y / 2 + yy) = (int16_t)tmp[yy * (w / 2) + xx];
        }
    };
    auto add_residual = [&](Plane &P, int x, int y, const int d[4][4]) {
        int r[4][4];
        itrans4x4(d, r);
        for (int yy = 0; yy < 4; ++yy) for (int xx = 0; xx < 4; ++xx) P.at(x + xx, y + yy) = (int16_t)clip8(P.at(x + xx, y + yy) + r[yy][xx]);
    };
    auto any = [](const int d[4][4]) { for (int y = 0; y < 4; ++y) for (int x = 0; x < 4; ++x) if (d[y][x]) return true; return false; };
    auto chroma_residual = [&](int mbx, int mby, int cbp_c, int qpc_) {
        int dcs[2][2][2] = {};
        int coef[16];
        for (int ci = 0; ci < 2; ++ci)
            if (cbp_c & 3) { residual_block(b, -1, 4, coef); chroma_dc(coef, qpc_, dcs[ci]); }
        for (int ci = 0; ci < 2; ++ci)
            for (int k = 0; k < 4; ++k) {
                const int cx = 2 * mbx + (k & 1), cy = 2 * mby + (k >> 1);
                int d[4][4] = {};
                if (cbp_c & 2) {
                    const int tot = residual_block(b, nc_of(nnzC[ci], C4w, cx, cy), 15, coef);
                    nnzC[ci][(size_t)cy * C4w + cx] = tot;
                    scale4x4(coef, 1, qpc_, true, d);
                }
                d[0][0] = dcs[ci][k >> 1][k & 1];
                if (any(d)) add_residual(pic.C[ci], 4 * cx, 4 * cy, d);
            }
    };
    auto intra_chroma_pred = [&](int mbx, int mby, int mode, bool left_ok, bool top_ok) {
        for (int ci = 0; ci < 2; ++ci) {
            Plane &P = pic.C[ci];
            const int x0 = 8 * mbx, y0 = 8 * mby;
            int top[8] = {0}, left[8] = {0};
            if (top_ok) for (int k = 0; k < 8; ++k) top[k] = P.at(x0 + k, y0 - 1);
            if (left_ok) for (int k = 0; k < 8; ++k) left[k] = P.at(x0 - 1, y0 + k);
            if ((mode == 1 && !left_ok) || (mode == 2 && !top_ok) || (mode == 3 && !(left_ok && top_ok)) || mode > 3) throw BadStream("intra chroma mode");
            if (mode == 0) {
                for (int k = 0; k < 4; ++k) {
                    const int xo = 4 * (k & 1), yo = 4 * (k >> 1);
                    const int st = top[xo] + top[xo + 1] + top[xo + 2] + top[xo + 3], sl = left[yo] + left[yo + 1] + left[yo + 2] + left[yo + 3];
                    int v;
                    if ((xo == 0 && yo == 0) || (xo == 4 && yo == 4)) v = (top_ok && left_ok) ? (st + sl + 4) >> 3 : (top_ok ? (st + 2) >> 2 : (left_ok ? (sl + 2) >> 2 : 128));
                    else if (xo == 4 && yo == 0) v = top_ok ? (st + 2) >> 2 : (left_ok ? (sl + 2) >> 2 : 128);
                    else v = left_ok ? (sl + 2) >> 2 : (top_ok ? (st + 2) >> 2 : 128);
                    for (int yy = 0; yy < 4; ++yy) for (int xx = 0; xx < 4; ++xx) P.at(x0 + xo + xx, y0 + yo + yy) = (int16_t)v;
                }
            } else if (mode == 1) { for (int yy = 0; yy < 8; ++yy) for (int xx = 0; xx < 8; ++xx) P.at(x0 + xx, y0 + yy) = (int16_t)left[yy]; }
            else if (mode == 2) { for (int yy = 0; yy < 8; ++yy) for (int xx = 0; xx < 8; ++xx) P.at(x0 + xx, y0 + yy) = (int16_t)top[xx]; }
            else {
                int out[64];
                pred_plane(top, left, P.at(x0 - 1, y0 - 1), 8, out);
                for (int yy = 0; yy < 8; ++yy) for (int xx = 0; xx < 8; ++xx) P.at(x0 + xx, y0 + yy) = (int16_t)out[yy * 8 + xx];
            }
        }
    };

    int mb = 0;
    int coef[16];
    for (;;) {
        if (is_p) {
            const unsigned run = b.ue();
            for (unsigned k = 0; k < run; ++k) {                    // P_Skip (8.4.1.1)
                if (mb >= nmb) throw BadStream("mb_skip_run past the picture");
                const int mbx = mb % mb_w, mby = mb / mb_w, bx = 4 * mbx, by = 4 * mby;
                const Nb A = mv_nb(bx - 1, by, mb), B = mv_nb(bx, by - 1, mb);
                MV mv{};
                if (!(!A.ok || !B.ok || (A.ref == 0 && A.mv == MV{}) || (B.ref == 0 && B.mv == MV{}))) mv = pred_mv(bx, by, 4, 0, mb, 0, 0);
                set_motion(bx, by, 4, 4, 0, mv);
                inter_pred(bx, by, 4, 4, 0, mv);
                mb_qp[mb] = qp;
                ++mb;
            }
            if (!b.more()) break;
        }
        if (mb >= nmb) throw BadStream("macroblock data past the picture");
        const int mbx = mb % mb_w, mby = mb / mb_w, bx0 = 4 * mbx, by0 = 4 * mby;
        const bool left_ok = mbx > 0, top_ok = mby > 0;
        const int mb_type = (int)b.ue();
        int intra_type = -1;
        if (is_p && mb_type >= 5) intra_type = mb_type - 5;
        else if (!is_p) intra_type = mb_type;
        if (intra_type >= 0) {
            if (intra_type > 25) throw BadStream("mb_type");
            mb_intra[mb] = 1;
            for (int yy = by0; yy < by0 + 4; ++yy) for (int xx = bx0; xx < bx0 + 4; ++xx) { refi[(size_t)yy * W4 + xx] = -1; refu[(size_t)yy * W4 + xx] = -1; mvs[(size_t)yy * W4 + xx] = MV{}; }
            if (intra_type == 25) {                                  // I_PCM
                b.align();
                for (int yy = 0; yy < 16; ++yy) for (int xx = 0; xx < 16; ++xx) Y.at(16 * mbx + xx, 16 * mby + yy) = (int16_t)b.u(8);
                for (int ci = 0; ci < 2; ++ci) for (int yy = 0; yy < 8; ++yy) for (int xx = 0; xx < 8; ++xx) pic.C[ci].at(8 * mbx + xx, 8 * mby + yy) = (int16_t)b.u(8);
                for (int yy = by0; yy < by0 + 4; ++yy) for (int xx = bx0; xx < bx0 + 4; ++xx) nnzY[(size_t)yy * W4 + xx] = 16;
                for (int ci = 0; ci < 2; ++ci) for (int yy = 2 * mby; yy < 2 * mby + 2; ++yy) for (int xx = 2 * mbx; xx < 2 * mbx + 2; ++xx) nnzC[ci][(size_t)yy * C4w + xx] = 16;
                mb_qp[mb] = 0;
                ++mb;
                if (!b.more()) break;
                continue;
            }
            int modes[16] = {0}, chroma_mode, cbp, i16mode = 0;
            if (intra_type == 0) {                                   // I_4x4 (8.3.1.1)
                for (int k = 0; k < 16; ++k) {
                    const int gx = bx0 + BLK_X[k], gy = by0 + BLK_Y[k];
                    auto nmode = [&](int nx, int ny) { if (nx < 0 || ny < 0) return -1; const int m = ipm[(size_t)ny * W4 + nx]; return m >= 0 ? m : 2; };
                    const int ma = nmode(gx - 1, gy), mbm = nmode(gx, gy - 1);
                    const int predm = (ma < 0 || mbm < 0) ? 2 : std::min(ma, mbm);
                    int mode;
                    if (b.bit()) mode = predm;
                    else { const int rem = (int)b.u(3); mode = rem < predm ? rem : rem + 1; }
                    ipm[(size_t)gy * W4 + gx] = mode;
                    modes[k] = mode;
                }
                chroma_mode = (int)b.ue();
                const unsigned c = b.ue();
                if (c > 47) throw BadStream("coded_block_pattern");
                cbp = CBP_INTRA[c];
            } else {
                i16mode = (intra_type - 1) % 4;
                cbp = ((((intra_type - 1) / 4) % 3) << 4) | (intra_type >= 13 ? 15 : 0);
                chroma_mode = (int)b.ue();
            }
            if (cbp > 0 || intra_type != 0) qp = next_qp(qp, b.se());
            mb_qp[mb] = qp;
            const int qpc_ = qpc_of(qp), x0 = 16 * mbx, y0 = 16 * mby;
            if (intra_type == 0) {
                for (int k = 0; k < 16; ++k) {
                    const int x4 = BLK_X[k], y4 = BLK_Y[k], gx = bx0 + x4, gy = by0 + y4, x = 4 * gx, y = 4 * gy;
                    const bool t_ok = gy > 0, l_ok = gx > 0;
                    int top[8], left[4], tl = 0;
                    if (t_ok) {
                        for (int i = 0; i < 4; ++i) top[i] = Y.at(x + i, y - 1);
                        bool tr_ok;
                        if (y4 == 0) tr_ok = gx + 1 < W4;
                        else {
                            tr_ok = false;
                            if (x4 < 3) for (int q = 0; q < k; ++q) if (BLK_X[q] == x4 + 1 && BLK_Y[q] == y4 - 1) tr_ok = true;
                        }
                        for (int i = 0; i < 4; ++i) top[4 + i] = tr_ok ? Y.at(x + 4 + i, y - 1) : top[3];
                    }
                    if (l_ok) for (int i = 0; i < 4; ++i) left[i] = Y.at(x - 1, y + i);
                    if (t_ok && l_ok) tl = Y.at(x - 1, y - 1);
                    int p[4][4];
                    pred_intra4x4(modes[k], t_ok ? top : nullptr, l_ok ? left : nullptr, tl, p);
                    for (int yy = 0; yy < 4; ++yy) for (int xx = 0; xx < 4; ++xx) Y.at(x + xx, y + yy) = (int16_t)p[yy][xx];
                    if (cbp & (1 << (2 * (y4 >> 1) + (x4 >> 1)))) {
                        const int tot = residual_block(b, nc_of(nnzY, W4, gx, gy), 16, coef);
                        nnzY[(size_t)gy * W4 + gx] = tot;
                        if (tot) { int d[4][4]; scale4x4(coef, 0, qp, false, d); add_residual(Y, x, y, d); }
                    }
                }
            } else {
                int top[16] = {0}, left[16] = {0};
                if (top_ok) for (int i = 0; i < 16; ++i) top[i] = Y.at(x0 + i, y0 - 1);
                if (left_ok) for (int i = 0; i < 16; ++i) left[i] = Y.at(x0 - 1, y0 + i);
                if ((i16mode == 0 && !top_ok) || (i16mode == 1 && !left_ok) || (i16mode == 3 && !(top_ok && left_ok))) throw BadStream("intra 16x16 mode");
                int out[256];
                if (i16mode == 3) pred_plane(top, left, Y.at(x0 - 1, y0 - 1), 16, out);
                else {
                    int st = 0, sl = 0;
                    for (int i = 0; i < 16; ++i) { st += top[i]; sl += left[i]; }
                    const int dcv = (top_ok && left_ok) ? (st + sl + 16) >> 5 : (top_ok ? (st + 8) >> 4 : (left_ok ? (sl + 8) >> 4 : 128));
                    for (int yy = 0; yy < 16; ++yy) for (int xx = 0; xx < 16; ++xx) out[yy * 16 + xx] = i16mode == 0 ? top[xx] : (i16mode == 1 ? left[yy] : dcv);
                }
                for (int yy = 0; yy < 16; ++yy) for (int xx = 0; xx < 16; ++xx) Y.at(x0 + xx, y0 + yy) = (int16_t)out[yy * 16 + xx];
                residual_block(b, nc_of(nnzY, W4, bx0, by0), 16, coef);
                int dc[4][4];
                luma_dc(coef, qp, dc);
                for (int k = 0; k < 16; ++k) {
                    const int x4 = BLK_X[k], y4 = BLK_Y[k], gx = bx0 + x4, gy = by0 + y4;
                    int d[4][4] = {};
                    if (cbp & 15) {
                        const int tot = residual_block(b, nc_of(nnzY, W4, gx, gy), 15, coef);
                        nnzY[(size_t)gy * W4 + gx] = tot;
                        scale4x4(coef, 1, qp, true, d);
                    }
                    d[0][0] = dc[y4][x4];
                    if (any(d)) add_residual(Y, 4 * gx, 4 * gy, d);
                }
            }
            intra_chroma_pred(mbx, mby, chroma_mode, left_ok, top_ok);
            chroma_residual(mbx, mby, cbp >> 4, qpc_);
        } else {
            // P macroblock (7.3.5.1 / 7.3.5.2, 8.4.1)
            for (int yy = by0; yy < by0 + 4; ++yy) for (int xx = bx0; xx < bx0 + 4; ++xx) refi[(size_t)yy * W4 + xx] = -2;
            const int nref = (int)reflist.size();
            struct Part { int x, y, w, h, shape, idx, ref; };
            std::vector<Part> parts;
            if (mb_type == 0) parts = {{0, 0, 4, 4, 0, 0, 0}};
            else if (mb_type == 1) parts = {{0, 0, 4, 2, 1, 0, 0}, {0, 2, 4, 2, 1, 1, 0}};
            else if (mb_type == 2) parts = {{0, 0, 2, 4, 2, 0, 0}, {2, 0, 2, 4, 2, 1, 0}};
            else if (mb_type == 3 || mb_type == 4) {
                int sub[4];
                for (int k = 0; k < 4; ++k) { sub[k] = (int)b.ue(); if (sub[k] > 3) throw BadStream("sub_mb_type"); }
                int refs4[4] = {0, 0, 0, 0};
                if (mb_type == 3 && h.num_ref > 1) for (int k = 0; k < 4; ++k) refs4[k] = b.te(h.num_ref - 1);
                for (int k = 0; k < 4; ++k) {
                    const int ox = 2 * (k & 1), oy = 2 * (k >> 1);
                    if (sub[k] == 0) parts.push_back({ox, oy, 2, 2, 0, 0, refs4[k]});
                    else if (sub[k] == 1) { parts.push_back({ox, oy, 2, 1, 0, 0, refs4[k]}); parts.push_back({ox, oy + 1, 2, 1, 0, 0, refs4[k]}); }
                    else if (sub[k] == 2) { parts.push_back({ox, oy, 1, 2, 0, 0, refs4[k]}); parts.push_back({ox + 1, oy, 1, 2, 0, 0, refs4[k]}); }
                    else for (int q = 0; q < 4; ++q) parts.push_back({ox + (q & 1), oy + (q >> 1), 1, 1, 0, 0, refs4[k]});
                }
            } else throw BadStream("mb_type");
            if (mb_type < 3 && h.num_ref > 1) for (auto &p : parts) p.ref = b.te(h.num_ref - 1);
            std::vector<MV> mvd(parts.size());
            for (auto &d : mvd) { d.x = b.se(); d.y = b.se(); }
            for (size_t i = 0; i < parts.size(); ++i) {
                const Part &p = parts[i];
                if (p.ref < 0 || p.ref >= nref) throw BadStream("ref_idx out of range");
                const MV pm = pred_mv(bx0 + p.x, by0 + p.y, p.w, p.ref, mb, p.shape, p.idx);
                const MV mv{pm.x + mvd[i].x, pm.y + mvd[i].y};
                set_motion(bx0 + p.x, by0 + p.y, p.w, p.h, p.ref, mv);
                inter_pred(bx0 + p.x, by0 + p.y, p.w, p.h, p.ref, mv);
            }
            const unsigned c = b.ue();
            if (c > 47) throw BadStream("coded_block_pattern");
            const int cbp = CBP_INTER[c];
            if (cbp > 0) qp = next_qp(qp, b.se());
            mb_qp[mb] = qp;
            for (int k = 0; k < 16; ++k) {
                const int x4 = BLK_X[k], y4 = BLK_Y[k];
                if (cbp & (1 << (2 * (y4 >> 1) + (x4 >> 1)))) {
                    const int gx = bx0 + x4, gy = by0 + y4;
                    const int tot = residual_block(b, nc_of(nnzY, W4, gx, gy), 16, coef);
                    nnzY[(size_t)gy * W4 + gx] = tot;
                    if (tot) { int d[4][4]; scale4x4(coef, 0, qp, false, d); add_residual(Y, 4 * gx, 4 * gy, d); }
                }
            }
            chroma_residual(mbx, mby, cbp >> 4, qpc_of(qp));
        }
        ++mb;
        if (!b.more()) break;
    }
    if (mb != nmb) throw BadStream("slice ends before the last macroblock");
    if (b.pos != b.end) throw BadStream("slice data does not end at the stop bit");
}

// ---------------------------------------------------------------------------------------------- deblocking filter (8.7)
void Decoder::deblock(Picture &pic, const SliceHdr &h)
{
    const int cqo = pps.chroma_qp_offset;
    auto qpc_of = [&](int q) { return (int)QPC[clip3(0, 51, q + cqo)]; };
    auto bs_of = [&](int px, int py, int qx, int qy, bool mb_edge, bool ip, bool iq) {
        if (ip || iq) return mb_edge ? 4 : 3;
        const size_t p = (size_t)py * W4 + px, q = (size_t)qy * W4 + qx;
        if (nnzY[p] || nnzY[q]) return 2;
        if (refu[p] != refu[q]) return 1;
        return (std::abs(mvs[p].x - mvs[q].x) >= 4 || std::abs(mvs[p].y - mvs[q].y) >= 4) ? 1 : 0;
    };
    // one line of samples across an edge: s[-1 - k] = p_k, s[k] = q_k through the accessor
    auto filt = [&](Plane &P, int x, int y, int dx, int dy, int bS, int qpav, bool luma) {
        if (!bS) return;
        const int ia = clip3(0, 51, qpav + h.alpha_off), ib = clip3(0, 51, qpav + h.beta_off);
        const int alpha = ALPHA[ia], beta = BETA[ib];
        auto S = [&](int k) -> int16_t & { return P.at(x + k * dx, y + k * dy); };   // k >= 0: q_k, k < 0: p_(-1-k)
        const int p0 = S(-1), p1 = S(-2), q0 = S(0), q1 = S(1);
        if (!(std::abs(p0 - q0) < alpha && std::abs(p1 - p0) < beta && std::abs(q1 - q0) < beta)) return;
        if (luma) {
            const int p2 = S(-3), q2 = S(2);
            const bool ap = std::abs(p2 - p0) < beta, aq = std::abs(q2 - q0) < beta;
            if (bS < 4) {
                const int tc0 = TC0[ia][bS - 1], tc = tc0 + ap + aq;
                const int delta = clip3(-tc, tc, ((q0 - p0) * 4 + (p1 - q1) + 4) >> 3);
                S(-1) = (int16_t)clip8(p0 + delta); S(0) = (int16_t)clip8(q0 - delta);
                if (ap) S(-2) = (int16_t)(p1 + clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - 2 * p1) >> 1));
                if (aq) S(1) = (int16_t)(q1 + clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - 2 * q1) >> 1));
            } else {
                const int p3 = S(-4), q3 = S(3);
                const bool small = std::abs(p0 - q0) < ((alpha >> 2) + 2);
                if (ap && small) { S(-1) = (int16_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3); S(-2) = (int16_t)((p2 + p1 + p0 + q0 + 2) >> 2); S(-3) = (int16_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3); }
                else S(-1) = (int16_t)((2 * p1 + p0 + q1 + 2) >> 2);
                if (aq && small) { S(0) = (int16_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3); S(1) = (int16_t)((p0 + q0 + q1 + q2 + 2) >> 2); S(2) = (int16_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3); }
                else S(0) = (int16_t)((2 * q1 + q0 + p1 + 2) >> 2);
            }
        } else if (bS < 4) {
            const int tc = TC0[ia][bS - 1] + 1;
            const int delta = clip3(-tc, tc, ((q0 - p0) * 4 + (p1 - q1) + 4) >> 3);
            S(-1) = (int16_t)clip8(p0 + delta); S(0) = (int16_t)clip8(q0 - delta);
        } else {
            S(-1) = (int16_t)((2 * p1 + p0 + q1 + 2) >> 2); S(0) = (int16_t)((2 * q1 + q0 + p1 + 2) >> 2);
        }
    };
    for (int mb = 0; mb < mb_w * mb_h; ++mb) {
        const int mbx = mb % mb_w, mby = mb / mb_w, bx0 = 4 * mbx, by0 = 4 * mby, x0 = 16 * mbx, y0 = 16 * mby;
        const bool iq = mb_intra[mb] != 0;
        for (int vertical = 1; vertical >= 0; --vertical)
            for (int e = 0; e < 4; ++e) {
                if (e == 0 && ((vertical && mbx == 0) || (!vertical && mby == 0))) continue;
                const int pmb = e == 0 ? (vertical ? mb - 1 : mb - mb_w) : mb;
                int bS4[4];
                bool anybs = false;
                for (int k = 0; k < 4; ++k) {
                    bS4[k] = vertical ? bs_of(bx0 + e - 1, by0 + k, bx0 + e, by0 + k, e == 0, mb_intra[pmb] != 0, iq)
                                      : bs_of(bx0 + k, by0 + e - 1, bx0 + k, by0 + e, e == 0, mb_intra[pmb] != 0, iq);
                    anybs |= bS4[k] != 0;
                }
                if (!anybs) continue;
                const int qpav = (mb_qp[pmb] + mb_qp[mb] + 1) >> 1;
                for (int i = 0; i < 16; ++i) {
                    if (vertical) filt(pic.Y, x0 + 4 * e, y0 + i, 1, 0, bS4[i >> 2], qpav, true);
                    else filt(pic.Y, x0 + i, y0 + 4 * e, 0, 1, bS4[i >> 2], qpav, true);
                }
                if (e % 2 == 0) {                                    // chroma edges 0 and 1 lie on luma edges 0 and 2
                    const int qpc_av = (qpc_of(mb_qp[pmb]) + qpc_of(mb_qp[mb]) + 1) >> 1;
                    for (int ci = 0; ci < 2; ++ci)
                        for (int i = 0; i < 8; ++i) {
                            if (vertical) filt(pic.C[ci], 8 * mbx + 2 * e, 8 * mby + i, 1, 0, bS4[i >> 1], qpc_av, false);
                            else filt(pic.C[ci], 8 * mbx + i, 8 * mby + 2 * e, 0, 1, bS4[i >> 1], qpc_av, false);
                        }
                }
            }
    }
}

std::shared_ptr<Picture> Decoder::decode_nal(const uint8_t *nal, size_t len)
{
    const int nal_type = nal[0] & 31, nal_ref_idc = (nal[0] >> 5) & 3;
    if (nal_type != 1 && nal_type != 5) return nullptr;
    Bits b(nal + 1, len - 1);
    const SliceHdr h = slice_header(b, nal_type, nal_ref_idc);
    if (h.idr) refs.clear();
    if (sps.mb_w < 1 || sps.mb_h < 1 || sps.mb_w > 512 || sps.mb_h > 512) throw BadStream("picture size out of range");   // (an in-band SPS)
    mb_w = sps.mb_w; mb_h = sps.mb_h; W4 = 4 * mb_w; H4 = 4 * mb_h;
    auto pic = std::make_shared<Picture>();
    pic->Y.init(16 * mb_w, 16 * mb_h);
    pic->C[0].init(8 * mb_w, 8 * mb_h); pic->C[1].init(8 * mb_w, 8 * mb_h);
    pic->frame_num = h.frame_num;
    pic->uid = npic++;
    nnzY.assign((size_t)W4 * H4, 0);
    for (int ci = 0; ci < 2; ++ci) nnzC[ci].assign((size_t)(W4 / 2) * (H4 / 2), 0);
    ipm.assign((size_t)W4 * H4, -1);
    refi.assign((size_t)W4 * H4, -2);
    refu.assign((size_t)W4 * H4, -1);
    mvs.assign((size_t)W4 * H4, MV{});
    mb_qp.assign((size_t)mb_w * mb_h, 0);
    mb_intra.assign((size_t)mb_w * mb_h, 0);
    decode_slice(b, h, *pic, h.type == 0 ? ref_list(h) : std::vector<std::shared_ptr<Picture>>());
    if (h.deblock != 1) deblock(*pic, h);
    if (nal_ref_idc) {
        const int max_fn = 1 << sps.log2_max_frame_num;
        for (auto &m : h.mmco) {                                     // MMCO 1: unmark a short-term picture
            const int want = h.frame_num - (m.second + 1);
            refs.erase(std::remove_if(refs.begin(), refs.end(), [&](const auto &p) { return (p->frame_num <= h.frame_num ? p->frame_num : p->frame_num - max_fn) == want; }), refs.end());
        }
        refs.insert(refs.begin(), pic);
        const size_t keep = (size_t)std::max(sps.num_ref_frames, 1);
        if (refs.size() > keep) refs.resize(keep);                    // sliding window (8.2.5.3)
    }
    return pic;
}

// ---------------------------------------------------------------------------------------------- ISO-BMFF sample tables
// every read of the file goes through a bounds check: a truncated or damaged file is an error status, never an out-of-range access
struct FileView {
    const std::vector<uint8_t> &f;
    uint8_t u8(size_t o) const { if (o >= f.size()) throw BadStream("box past the end of the file"); return f[o]; }
    uint32_t u32(size_t o) const { if (o + 4 > f.size() || o + 4 < o) throw BadStream("box past the end of the file"); return ((uint32_t)f[o] << 24) | ((uint32_t)f[o + 1] << 16) | ((uint32_t)f[o + 2] << 8) | f[o + 3]; }
    uint64_t u64(size_t o) const { return ((uint64_t)u32(o) << 32) | u32(o + 4); }
    bool is(size_t o, const char *four) const { return o + 4 <= f.size() && std::memcmp(&f[o], four, 4) == 0; }
};

struct Box { size_t lo, hi; };
bool find_box(const FileView &fv, size_t lo, size_t hi, const char *const *path, int depth, Box &out)
{
    hi = std::min(hi, fv.f.size());
    while (lo + 8 <= hi) {
        uint64_t size = fv.u32(lo);
        size_t hdr = 8;
        if (size == 1) { if (lo + 16 > hi) return false; size = fv.u64(lo + 8); hdr = 16; }
        else if (size == 0) size = hi - lo;
        if (size < hdr || size > hi - lo) return false;
        if (fv.is(lo + 4, path[0])) {
            if (depth == 1) { out = Box{lo + hdr, lo + (size_t)size}; return true; }
            if (find_box(fv, lo + hdr, lo + (size_t)size, path + 1, depth - 1, out)) return true;
        }
        lo += (size_t)size;
    }
    return false;
}

}  // namespace

struct lkingest {
    std::vector<uint8_t> file;
    std::vector<std::pair<size_t, size_t>> samples;   // offset, size
    int nls = 4;
    size_t next = 0;
    Decoder dec;
    std::string err;
    std::shared_ptr<Picture> last;
};

namespace {

void open_tables(lkingest &v)
{
    const FileView fv{v.file};
    Box moov;
    const char *pm[] = {"moov"};
    if (!find_box(fv, 0, v.file.size(), pm, 1, moov)) throw BadStream("no moov box");
    size_t lo = moov.lo;
    while (lo + 8 <= moov.hi) {
        const uint64_t size = fv.u32(lo);
        if (size < 8 || size > moov.hi - lo) break;
        if (fv.is(lo + 4, "trak")) {
            Box stbl, stsd, b2;
            const char *ps[] = {"mdia", "minf", "stbl"};
            const char *p1[] = {"stsd"}, *p2[] = {"stsz"}, *p3[] = {"stco"}, *p4[] = {"co64"}, *p5[] = {"stsc"}, *p6[] = {"avcC"};
            if (find_box(fv, lo + 8, lo + (size_t)size, ps, 3, stbl) && find_box(fv, stbl.lo, stbl.hi, p1, 1, stsd) && fv.is(stsd.lo + 12, "avc1")) {
                const size_t entry = stsd.lo + 8, esz = fv.u32(entry);
                Box avcc;
                if (esz < 86 || esz > stsd.hi - entry || !find_box(fv, entry + 8 + 78, entry + esz, p6, 1, avcc)) throw BadStream("no avcC box");
                // AVCDecoderConfigurationRecord: every field inside the box
                auto c = [&](size_t p) { if (avcc.lo + p >= avcc.hi) throw BadStream("avcC too short"); return fv.u8(avcc.lo + p); };
                v.nls = (c(4) & 3) + 1;
                size_t p = 6;
                for (int k = c(5) & 31; k > 0; --k) {
                    const size_t n = ((size_t)c(p) << 8) | c(p + 1);
                    if (n < 4 || avcc.lo + p + 2 + n > avcc.hi) throw BadStream("avcC: bad parameter-set length");
                    v.dec.sps = parse_sps(&v.file[avcc.lo + p + 2], n);
                    p += 2 + n;
                }
                int npps = c(p++);
                for (; npps > 0; --npps) {
                    const size_t n = ((size_t)c(p) << 8) | c(p + 1);
                    if (n < 2 || avcc.lo + p + 2 + n > avcc.hi) throw BadStream("avcC: bad parameter-set length");
                    v.dec.pps = parse_pps(&v.file[avcc.lo + p + 2], n);
                    p += 2 + n;
                }
                if (v.dec.sps.mb_w < 1 || v.dec.sps.mb_h < 1 || v.dec.sps.mb_w > 512 || v.dec.sps.mb_h > 512) throw BadStream("picture size out of range");
                if (!find_box(fv, stbl.lo, stbl.hi, p2, 1, b2)) throw BadStream("no stsz box");
                const uint32_t fixed = fv.u32(b2.lo + 4), count = fv.u32(b2.lo + 8);
                if (count > (1u << 24) || (!fixed && (size_t)count * 4 > b2.hi - b2.lo)) throw BadStream("stsz: bad sample count");
                std::vector<uint32_t> sizes(count, fixed);
                if (!fixed) for (uint32_t k = 0; k < count; ++k) sizes[k] = fv.u32(b2.lo + 12 + 4 * (size_t)k);
                std::vector<uint64_t> chunks;
                if (find_box(fv, stbl.lo, stbl.hi, p3, 1, b2)) {
                    const uint32_t n = fv.u32(b2.lo + 4);
                    if ((size_t)n * 4 > b2.hi - b2.lo) throw BadStream("stco: bad entry count");
                    for (uint32_t k = 0; k < n; ++k) chunks.push_back(fv.u32(b2.lo + 8 + 4 * (size_t)k));
                } else if (find_box(fv, stbl.lo, stbl.hi, p4, 1, b2)) {
                    const uint32_t n = fv.u32(b2.lo + 4);
                    if ((size_t)n * 8 > b2.hi - b2.lo) throw BadStream("co64: bad entry count");
                    for (uint32_t k = 0; k < n; ++k) chunks.push_back(fv.u64(b2.lo + 8 + 8 * (size_t)k));
                } else throw BadStream("no chunk offsets");
                if (!find_box(fv, stbl.lo, stbl.hi, p5, 1, b2)) throw BadStream("no stsc box");
                const uint32_t nsc = fv.u32(b2.lo + 4);
                if ((size_t)nsc * 12 > b2.hi - b2.lo) throw BadStream("stsc: bad entry count");
                std::vector<std::pair<uint32_t, uint32_t>> stsc;
                for (uint32_t s = 0; s < nsc; ++s) stsc.push_back({fv.u32(b2.lo + 8 + 12 * (size_t)s), fv.u32(b2.lo + 12 + 12 * (size_t)s)});
                size_t k = 0;
                for (size_t ci = 0; ci < chunks.size() && k < count; ++ci) {
                    uint32_t per = 0;
                    for (const auto &e : stsc) if (ci + 1 >= e.first) per = e.second;
                    uint64_t off = chunks[ci];
                    for (uint32_t s = 0; s < per && k < count; ++s, ++k) {
                        if (off > v.file.size() || sizes[k] > v.file.size() - off) throw BadStream("sample outside the file");
                        v.samples.push_back({(size_t)off, sizes[k]});
                        off += sizes[k];
                    }
                }
                return;
            }
        }
        lo += (size_t)size;
    }
    throw BadStream("no avc1 track");
}

int read_picture(lkingest *v)      // 1: v->last holds a new picture, 0: end of stream
{
    while (v->next < v->samples.size()) {
        const size_t off = v->samples[v->next].first, size = v->samples[v->next].second;
        ++v->next;
        std::shared_ptr<Picture> pic;
        size_t p = 0;
        while (p + v->nls <= size) {
            size_t n = 0;
            for (int k = 0; k < v->nls; ++k) n = (n << 8) | v->file[off + p + k];
            p += v->nls;
            if (n == 0 || p + n > size) throw BadStream("NAL unit outside its sample");
            const uint8_t *nal = &v->file[off + p];
            const int t = nal[0] & 31;
            if (t == 7) v->dec.sps = parse_sps(nal, n);
            else if (t == 8) v->dec.pps = parse_pps(nal, n);
            else if (auto q = v->dec.decode_nal(nal, n)) pic = q;
            p += n;
        }
        if (pic) { v->last = pic; return 1; }
    }
    return 0;
}

template <typename F>
int guarded(lkingest *v, F &&f)
{
    try { return f(); }
    catch (const Unsupported &e) { v->err = e.what(); return LKINGEST_E_UNSUPPORTED; }
    catch (const BadStream &e) { v->err = e.what(); return LKINGEST_E_STREAM; }
    catch (const std::exception &e) { v->err = e.what(); return LKINGEST_E_STREAM; }
}

}  // namespace

extern "C" {

int lkingest_open(lkingest **out, const char *path)
{
    if (!out || !path) return LKINGEST_E_ARG;
    *out = nullptr;
    std::unique_ptr<lkingest> v(new (std::nothrow) lkingest());
    if (!v) return LKINGEST_E_IO;
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return LKINGEST_E_IO;
    std::fseek(fp, 0, SEEK_END);
    const long sz = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    if (sz <= 0) { std::fclose(fp); return LKINGEST_E_IO; }
    v->file.resize((size_t)sz);
    const size_t got = std::fread(v->file.data(), 1, (size_t)sz, fp);
    std::fclose(fp);
    if (got != (size_t)sz) return LKINGEST_E_IO;
    const int rc = guarded(v.get(), [&] { open_tables(*v); return LKINGEST_OK; });
    if (rc == LKINGEST_E_STREAM) return LKINGEST_E_IO;       // not an mp4 this reader understands
    if (rc) return rc;
    *out = v.release();
    return LKINGEST_OK;
}

int lkingest_info(const lkingest *v, int *rows, int *cols, int *n_frames)
{
    if (!v) return LKINGEST_E_ARG;
    const SPS &s = v->dec.sps;
    if (rows) *rows = 16 * s.mb_h - 2 * (s.crop[2] + s.crop[3]);
    if (cols) *cols = 16 * s.mb_w - 2 * (s.crop[0] + s.crop[1]);
    if (n_frames) *n_frames = (int)v->samples.size();
    return LKINGEST_OK;
}

int lkingest_read_yuv(lkingest *v, unsigned char *y, size_t y_step, unsigned char *cb, unsigned char *cr, size_t c_step)
{
    if (!v) return LKINGEST_E_ARG;
    int rows, cols;
    lkingest_info(v, &rows, &cols, nullptr);
    if ((y && y_step < (size_t)cols) || ((cb || cr) && c_step < (size_t)cols / 2)) return LKINGEST_E_ARG;
    return guarded(v, [&] {
        const int got = read_picture(v);
        if (got != 1) return got;
        const SPS &s = v->dec.sps;
        const Picture &p = *v->last;
        if (y) for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) y[(size_t)r * y_step + c] = (unsigned char)p.Y.p[(size_t)(r + 2 * s.crop[2]) * p.Y.w + c + 2 * s.crop[0]];
        unsigned char *dst[2] = {cb, cr};
        for (int ci = 0; ci < 2; ++ci)
            if (dst[ci]) for (int r = 0; r < rows / 2; ++r) for (int c = 0; c < cols / 2; ++c) dst[ci][(size_t)r * c_step + c] = (unsigned char)p.C[ci].p[(size_t)(r + s.crop[2]) * p.C[ci].w + c + s.crop[0]];
        return 1;
    });
}

int lkingest_read_bgr(lkingest *v, unsigned char *bgr, size_t step)
{
    if (!v || !bgr) return LKINGEST_E_ARG;
    int rows, cols;
    lkingest_info(v, &rows, &cols, nullptr);
    if (step < (size_t)cols * 3) return LKINGEST_E_ARG;
    return guarded(v, [&] {
        const int got = read_picture(v);
        if (got != 1) return got;
        const SPS &s = v->dec.sps;
        const Picture &p = *v->last;
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) {
                const int yy = r + 2 * s.crop[2], xx = c + 2 * s.crop[0];
                const int C = p.Y.p[(size_t)yy * p.Y.w + xx] - 16, D = p.C[0].p[(size_t)(yy / 2) * p.C[0].w + xx / 2] - 128, E = p.C[1].p[(size_t)(yy / 2) * p.C[1].w + xx / 2] - 128;
                unsigned char *o = bgr + (size_t)r * step + 3 * (size_t)c;
                o[0] = (unsigned char)clip8((298 * C + 516 * D + 128) >> 8);
                o[1] = (unsigned char)clip8((298 * C - 100 * D - 208 * E + 128) >> 8);
                o[2] = (unsigned char)clip8((298 * C + 409 * E + 128) >> 8);
            }
        return 1;
    });
}

void lkingest_close(lkingest *v) { delete v; }
const char *lkingest_last_error(const lkingest *v) { return v ? v->err.c_str() : ""; }

}  // extern "C"
