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
    if (W <= 0 || H <= 0 || W > LZ_IMAGE_MAX_DIM || H > LZ_IMAGE_MAX_DIM || depth != 8 || interlace != 0 || !z) {
        g_err = "unsupported PNG (need 8-bit, non-interlaced)";
        free(z);
        return NULL;
    }
    int sc = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!sc) {
        g_err = "unsupported PNG colour type";
        free(z);
        return NULL;
    }
    size_t stride = (size_t)W * sc;
    uLongf rawlen = (uLongf)((stride + 1) * H);
    uint8_t* raw = (uint8_t*)malloc(rawlen);
    if (!raw || uncompress(raw, &rawlen, z, (uLong)zlen) != Z_OK || rawlen != (stride + 1) * H) {
        g_err = "PNG inflate failed";
        free(raw);
        free(z);
        return NULL;
    }
    free(z);
    uint8_t* img = (uint8_t*)malloc(stride * H);
    if (!img) {
        free(raw);
        return NULL;
    }
    for (int y = 0; y < H; y++) {
        const uint8_t* s = raw + (stride + 1) * y;
        uint8_t* d = img + stride * y;
        const uint8_t* up = y ? d - stride : NULL;
        int ft = s[0];
        s++;
        for (size_t x = 0; x < stride; x++) {
            int a = x >= (size_t)sc ? d[x - sc] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)sc) ? up[x - sc] : 0;
            int v = s[x];
            switch (ft) {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: break;
            }
            d[x] = (uint8_t)v;
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

uint8_t* lz_image_load(const char* path, int* w, int* h, int* channels_in_file, int desired_channels) {
    size_t n = 0;
    uint8_t* buf = read_file(path, &n);
    if (!buf) {
        g_err = "cannot read file";
        return NULL;
    }
    int W = 0, H = 0, sc = 0;
    uint8_t* img = (n > 2 && buf[0] == 'P') ? load_pnm(buf, n, &W, &H, &sc) : load_png(buf, n, &W, &H, &sc);
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
