/* image_io.c -- see image_io.h.  PNG through zlib (inflate/deflate + the five scanline filters), PNM. */
#include "image_io.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* largest width / height accepted from a file header (stb's STBI_MAX_DIMENSIONS is 1 << 24) */
#define LZ_IMAGE_MAX_DIM (1 << 24)

static const char* g_err = "";
const char* lz_image_last_error(void) { return g_err; }
void lz_image_free(void* p) { free(p); }

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }
static void put32(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}

static uint8_t* read_file(const char* path, size_t* n) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz <= 0) {
        fclose(f);
        return NULL;
    }
    uint8_t* b = (uint8_t*)malloc((size_t)sz);
    if (b && fread(b, 1, (size_t)sz, f) != (size_t)sz) {
        free(b);
        b = NULL;
    }
    fclose(f);
    *n = (size_t)sz;
    return b;
}

/* convert `src` (w*h pixels, sc channels) to dc channels, stb conventions (gray = luma, alpha = 255) */
static uint8_t* convert_channels(uint8_t* src, int w, int h, int sc, int dc) {
    if (sc == dc) return src;
    uint8_t* dst = (uint8_t*)malloc((size_t)w * h * dc);
    if (!dst) {
        free(src);
        return NULL;
    }
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint8_t* s = src + i * sc;
        uint8_t r, g, b, a = 255;
        if (sc <= 2) {
            r = g = b = s[0];
            if (sc == 2) a = s[1];
        } else {
            r = s[0];
            g = s[1];
            b = s[2];
            if (sc == 4) a = s[3];
        }
        uint8_t* d = dst + i * dc;
        if (dc <= 2) {
            d[0] = (uint8_t)((r * 77 + g * 150 + b * 29) >> 8);
            if (dc == 2) d[1] = a;
        } else {
            d[0] = r;
            d[1] = g;
            d[2] = b;
            if (dc == 4) d[3] = a;
        }
    }
    free(src);
    return dst;
}

static int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

static uint8_t* load_png(const uint8_t* buf, size_t n, int* w, int* h, int* ch) {
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (n < 33 || memcmp(buf, sig, 8)) {
        g_err = "not a PNG";
        return NULL;
    }
    size_t pos = 8, zcap = 0, zlen = 0;
    uint8_t* z = NULL;
    int W = 0, H = 0, depth = 0, ctype = 0, interlace = 0;
    uint8_t pal[256][4];
    int npal = 0;
    memset(pal, 255, sizeof(pal));
    while (pos + 12 <= n) {
        uint32_t len = be32(buf + pos);
        const uint8_t* type = buf + pos + 4;
        const uint8_t* data = buf + pos + 8;
        if (pos + 12 + len > n) break;
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            W = (int)be32(data);
            H = (int)be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            npal = (int)(len / 3);
            for (int i = 0; i < npal && i < 256; i++) memcpy(pal[i], data + 3 * i, 3);
        } else if (!memcmp(type, "tRNS", 4) && ctype == 3) {
            for (uint32_t i = 0; i < len && i < 256; i++) pal[i][3] = data[i];
        } else if (!memcmp(type, "IDAT", 4)) {
            if (zlen + len > zcap) {
                zcap = (zlen + len) * 2;
                z = (uint8_t*)realloc(z, zcap);
                if (!z) return NULL;
            }
            memcpy(z + zlen, data, len);
            zlen += len;
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    const int depth_ok = depth == 8 || depth == 16 || ((depth == 1 || depth == 2 || depth == 4) && (ctype == 0 || ctype == 3));
    if (W <= 0 || H <= 0 || W > LZ_IMAGE_MAX_DIM || H > LZ_IMAGE_MAX_DIM || !depth_ok || interlace > 1 || !z ||
        (ctype == 3 && depth == 16)) {
        g_err = "unsupported PNG (bit depth / interlace method)";
        free(z);
        return NULL;
    }
    int sc = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!sc) {
        g_err = "unsupported PNG colour type";
        free(z);
        return NULL;
    }
    /* One pass (non-interlaced) or the seven Adam7 passes: each is a reduced image with its own filtered scanlines. */
    static const int ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1};
    static const int adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
    const int npass = interlace ? 7 : 1;
    const int bpp_bits = sc * depth;                  /* bits per pixel in the file */
    const int fbpp = (bpp_bits + 7) / 8;              /* filter distance in bytes */
    size_t rawlen_need = 0;
    for (int p = 0; p < npass; p++) {
        const int pw = interlace ? (W - ax0[p] + adx[p] - 1) / adx[p] : W, ph = interlace ? (H - ay0[p] + ady[p] - 1) / ady[p] : H;
        if (pw > 0 && ph > 0) rawlen_need += ((size_t)((size_t)pw * bpp_bits + 7) / 8 + 1) * ph;
    }
    uLongf rawlen = (uLongf)rawlen_need;
    uint8_t* raw = (uint8_t*)malloc(rawlen_need ? rawlen_need : 1);
    if (!raw || uncompress(raw, &rawlen, z, (uLong)zlen) != Z_OK || rawlen != rawlen_need) {
        g_err = "PNG inflate failed";
        free(raw);
        free(z);
        return NULL;
    }
    free(z);
    uint8_t* img = (uint8_t*)malloc((size_t)W * H * sc); /* 8 bits per sample: 16-bit files keep their high byte (stb) */
    if (!img) {
        free(raw);
        return NULL;
    }
    size_t rpos = 0;
    for (int p = 0; p < npass; p++) {
        const int pw = interlace ? (W - ax0[p] + adx[p] - 1) / adx[p] : W, ph = interlace ? (H - ay0[p] + ady[p] - 1) / ady[p] : H;
        if (pw <= 0 || ph <= 0) continue;
        const size_t rb = ((size_t)pw * bpp_bits + 7) / 8; /* bytes of one scanline of this pass */
        uint8_t* prev = NULL;
        for (int y = 0; y < ph; y++) {
            uint8_t* line = raw + rpos;
            const int ft = line[0];
            line++;
            for (size_t x = 0; x < rb; x++) { /* un-filter in place */
                const int a = x >= (size_t)fbpp ? line[x - fbpp] : 0, b = prev ? prev[x] : 0, c = (prev && x >= (size_t)fbpp) ? prev[x - fbpp] : 0;
                int v = line[x];
                switch (ft) {
                    case 1: v += a; break;
                    case 2: v += b; break;
                    case 3: v += (a + b) >> 1; break;
                    case 4: v += paeth(a, b, c); break;
                    default: break;
                }
                line[x] = (uint8_t)v;
            }
            const int oy = interlace ? ay0[p] + y * ady[p] : y;
            for (int x = 0; x < pw; x++) {
                const int ox = interlace ? ax0[p] + x * adx[p] : x;
                uint8_t* d = img + ((size_t)oy * W + ox) * sc;
                for (int k = 0; k < sc; k++) {
                    if (depth == 8) d[k] = line[(size_t)x * sc + k];
                    else if (depth == 16) d[k] = line[((size_t)x * sc + k) * 2];
                    else { /* 1, 2, 4 bits: one channel; gray is scaled to 0..255, palette indices stay */
                        const int bit = x * depth, v = (line[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
                        d[k] = (uint8_t)(ctype == 3 ? v : v * 255 / ((1 << depth) - 1));
                    }
                }
            }
            prev = line;
            rpos += rb + 1;
        }
    }
    free(raw);
    if (ctype == 3) { /* expand the palette to RGBA */
        uint8_t* e = (uint8_t*)malloc((size_t)W * H * 4);
        if (!e) {
            free(img);
            return NULL;
        }
        for (size_t i = 0; i < (size_t)W * H; i++) memcpy(e + 4 * i, pal[img[i]], 4);
        free(img);
        img = e;
        sc = 4;
    }
    *w = W;
    *h = H;
    *ch = sc;
    return img;
}

static uint8_t* load_pnm(const uint8_t* buf, size_t n, int* w, int* h, int* ch) {
    if (n < 7 || buf[0] != 'P' || (buf[1] != '5' && buf[1] != '6')) return NULL;
    size_t pos = 2;
    int vals[3], got = 0;
    while (got < 3 && pos < n) {
        while (pos < n && (buf[pos] == ' ' || buf[pos] == '\n' || buf[pos] == '\r' || buf[pos] == '\t')) pos++;
        if (pos < n && buf[pos] == '#') {
            while (pos < n && buf[pos] != '\n') pos++;
            continue;
        }
        int v = 0, any = 0;
        while (pos < n && buf[pos] >= '0' && buf[pos] <= '9') {
            if (v > LZ_IMAGE_MAX_DIM) { /* header fields are bounded: no signed overflow on hostile input */
                g_err = "PNM header value too large";
                return NULL;
            }
            v = v * 10 + (buf[pos++] - '0');
            any = 1;
        }
        if (!any) return NULL;
        vals[got++] = v;
    }
    pos++; /* single whitespace after maxval */
    int c = buf[1] == '6' ? 3 : 1;
    if (got < 3 || vals[0] <= 0 || vals[1] <= 0 || vals[0] > LZ_IMAGE_MAX_DIM || vals[1] > LZ_IMAGE_MAX_DIM ||
        vals[2] != 255 || pos > n || (size_t)vals[0] * vals[1] * c > n - pos) {
        g_err = "unsupported PNM (need maxval 255)";
        return NULL;
    }
    uint8_t* img = (uint8_t*)malloc((size_t)vals[0] * vals[1] * c);
    if (!img) return NULL;
    memcpy(img, buf + pos, (size_t)vals[0] * vals[1] * c);
    *w = vals[0];
    *h = vals[1];
    *ch = c;
    return img;
}

/* Windows BMP: BITMAPINFOHEADER and later, uncompressed 24 / 32 bits per pixel (BI_RGB, BI_BITFIELDS with byte-aligned
 * masks) and 8-bit palettes; bottom-up or top-down. */
static uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint8_t* load_bmp(const uint8_t* buf, size_t n, int* w, int* h, int* ch) {
    if (n < 54 || buf[0] != 'B' || buf[1] != 'M') return NULL;
    const uint32_t off = le32(buf + 10), hdr = le32(buf + 14);
    if (hdr < 40 || 14 + (size_t)hdr > n) {
        g_err = "unsupported BMP header";
        return NULL;
    }
    const int32_t W = (int32_t)le32(buf + 18), Hs = (int32_t)le32(buf + 22);
    const int bpp = buf[28] | (buf[29] << 8);
    const uint32_t comp = le32(buf + 30);
    const int H = Hs < 0 ? -Hs : Hs, flip = Hs > 0;
    if (W <= 0 || H <= 0 || W > LZ_IMAGE_MAX_DIM || H > LZ_IMAGE_MAX_DIM || (bpp != 8 && bpp != 24 && bpp != 32) ||
        (comp != 0 && !(comp == 3 && bpp == 32))) {
        g_err = "unsupported BMP (need uncompressed 8 / 24 / 32 bpp)";
        return NULL;
    }
    int shift[4] = {16, 8, 0, 24}; /* BGRA byte order as R, G, B, A shifts */
    int has_alpha = 0;
    if (comp == 3 && (size_t)14 + 40 + 12 <= n) {
        for (int k = 0; k < 3 + (hdr >= 56); k++) {
            const uint32_t m = le32(buf + 54 + 4 * k);
            if (m != 0xffu && m != 0xff00u && m != 0xff0000u && m != 0xff000000u) {
                if (k == 3 && m == 0) continue;
                g_err = "unsupported BMP bit masks";
                return NULL;
            }
            shift[k] = m == 0xffu ? 0 : m == 0xff00u ? 8 : m == 0xff0000u ? 16 : 24;
            if (k == 3) has_alpha = 1;
        }
    }
    const size_t stride = (((size_t)W * bpp + 31) / 32) * 4;
    if (off > n || stride * H > n - off) {
        g_err = "truncated BMP";
        return NULL;
    }
    const uint8_t* pal = buf + 14 + hdr;
    uint32_t ncol = le32(buf + 46);
    if (bpp == 8) {
        if (ncol == 0 || ncol > 256) ncol = 256;
        if ((size_t)14 + hdr + 4 * (size_t)ncol > n) {
            g_err = "truncated BMP palette";
            return NULL;
        }
    }
    const int sc = (bpp == 32 && has_alpha) ? 4 : 3;
    uint8_t* img = (uint8_t*)malloc((size_t)W * H * sc);
    if (!img) return NULL;
    for (int y = 0; y < H; y++) {
        const uint8_t* s = buf + off + stride * (size_t)(flip ? H - 1 - y : y);
        uint8_t* d = img + (size_t)y * W * sc;
        for (int x = 0; x < W; x++) {
            if (bpp == 8) {
                const uint8_t* e = pal + 4 * (s[x] < ncol ? s[x] : 0);
                d[3 * x] = e[2];
                d[3 * x + 1] = e[1];
                d[3 * x + 2] = e[0];
            } else if (bpp == 24) {
                d[3 * x] = s[3 * x + 2];
                d[3 * x + 1] = s[3 * x + 1];
                d[3 * x + 2] = s[3 * x];
            } else {
                const uint32_t v = le32(s + 4 * x);
                for (int k = 0; k < sc; k++) d[sc * x + k] = (uint8_t)(v >> shift[k]);
            }
        }
    }
    *w = W;
    *h = H;
    *ch = sc;
    return img;
}

#ifdef LANCZOS_WITH_STB
/* Built with -DLANCZOS_WITH_STB -I<directory holding stb_image.h / stb_image_write.h> (make harness_stb STB_DIR=...): the
 * reference's own decoder and encoder (stb.cpp:1-6, full_TB.h:107,172) behind the same four functions -- every format
 * stbi_load accepts, including JPEG.  The headers are NOT part of this repository; a maintainer who drops the library into
 * the reference tree points STB_DIR at LanczosUpscaler/stb_image. */
#define STBI_NO_SIMD
#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include "stb_image_write.h"
#endif

uint8_t* lz_image_load(const char* path, int* w, int* h, int* channels_in_file, int desired_channels) {
#ifdef LANCZOS_WITH_STB
    int sw = 0, sh = 0, sc2 = 0;
    uint8_t* p = stbi_load(path, &sw, &sh, &sc2, desired_channels);
    if (!p) {
        g_err = "stbi_load failed";
        return NULL;
    }
    if (w) *w = sw;
    if (h) *h = sh;
    if (channels_in_file) *channels_in_file = sc2;
    return p;
#endif
    size_t n = 0;
    uint8_t* buf = read_file(path, &n);
    if (!buf) {
        g_err = "cannot read file";
        return NULL;
    }
    int W = 0, H = 0, sc = 0;
    uint8_t* img = (n > 2 && buf[0] == 'P') ? load_pnm(buf, n, &W, &H, &sc)
                   : (n > 2 && buf[0] == 'B' && buf[1] == 'M') ? load_bmp(buf, n, &W, &H, &sc)
                                                                : load_png(buf, n, &W, &H, &sc);
    free(buf);
    if (!img) return NULL;
    if (w) *w = W;
    if (h) *h = H;
    if (channels_in_file) *channels_in_file = sc;
    if (desired_channels > 0 && desired_channels != sc) img = convert_channels(img, W, H, sc, desired_channels);
    return img;
}

static void write_chunk(FILE* f, const char* type, const uint8_t* data, uint32_t len) {
    uint8_t hdr[8];
    put32(hdr, len);
    memcpy(hdr + 4, type, 4);
    fwrite(hdr, 1, 8, f);
    if (len) fwrite(data, 1, len, f);
    uLong crc = crc32(0L, (const Bytef*)type, 4);
    if (len) crc = crc32(crc, data, len);
    uint8_t c[4];
    put32(c, (uint32_t)crc);
    fwrite(c, 1, 4, f);
}

int lz_image_write_png(const char* path, int w, int h, int comp, const void* data, int stride_bytes) {
#ifdef LANCZOS_WITH_STB
    return stbi_write_png(path, w, h, comp, data, stride_bytes ? stride_bytes : w * comp);
#endif
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    static const int ctype_of[5] = {0, 0, 4, 2, 6};
    if (w <= 0 || h <= 0 || comp < 1 || comp > 4 || !data) return 0;
    if (stride_bytes == 0) stride_bytes = w * comp;
    size_t row = (size_t)w * comp;
    uint8_t* raw = (uint8_t*)malloc((row + 1) * h);
    if (!raw) return 0;
    for (int y = 0; y < h; y++) {
        raw[(row + 1) * y] = 0; /* filter type None */
        memcpy(raw + (row + 1) * y + 1, (const uint8_t*)data + (size_t)stride_bytes * y, row);
    }
    uLongf zlen = compressBound((uLong)((row + 1) * h));
    uint8_t* z = (uint8_t*)malloc(zlen);
    if (!z || compress2(z, &zlen, raw, (uLong)((row + 1) * h), 6) != Z_OK) {
        free(raw);
        free(z);
        return 0;
    }
    free(raw);
    FILE* f = fopen(path, "wb");
    if (!f) {
        free(z);
        return 0;
    }
    fwrite(sig, 1, 8, f);
    uint8_t ihdr[13];
    put32(ihdr, (uint32_t)w);
    put32(ihdr + 4, (uint32_t)h);
    ihdr[8] = 8;
    ihdr[9] = (uint8_t)ctype_of[comp];
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    write_chunk(f, "IHDR", ihdr, 13);
    write_chunk(f, "IDAT", z, (uint32_t)zlen);
    write_chunk(f, "IEND", NULL, 0);
    free(z);
    return fclose(f) == 0;
}

int lz_image_write_pnm(const char* path, int w, int h, int comp, const void* data, int stride_bytes) {
    if (w <= 0 || h <= 0 || (comp != 1 && comp != 3) || !data) return 0;
    if (stride_bytes == 0) stride_bytes = w * comp;
    FILE* f = fopen(path, "wb");
    if (!f) return 0;
    fprintf(f, "P%c\n%d %d\n255\n", comp == 3 ? '6' : '5', w, h);
    for (int y = 0; y < h; y++) fwrite((const uint8_t*)data + (size_t)stride_bytes * y, 1, (size_t)w * comp, f);
    return fclose(f) == 0;
}
