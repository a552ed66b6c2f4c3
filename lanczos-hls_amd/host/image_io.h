/*
 * image_io.h -- interleaved 8-bit image load/store for the harness.
 *
 * The reference's harness uses stb_image / stb_image_write (stb.cpp, full_TB.h:107,172): stbi_load returns
 * interleaved u8 HWC with the requested channel count; stbi_write_png takes the same layout.  This is our own
 * small codec with the SAME buffer contract (so a caller that already links stb can keep using it and just
 * hand the buffers to lanczos_u8): PNG through zlib (1/2/4/8/16 bits per sample, gray / gray+alpha / RGB / RGBA / palette,
 * Adam7 interlace; 16-bit samples keep their high byte, as stb does), BMP (uncompressed 8 / 24 / 32 bpp) and binary PPM/PGM.
 * NOT decoded by the own codec: JPEG, GIF, PSD, TGA, HDR, PIC (stbi_load takes them).  For those -- or to run on exactly
 * the reference's decoder -- build with the reference's stb headers where they lie: `make -C lanczos-hls_amd harness_stb
 * STB_DIR=<reference>/LanczosUpscaler/stb_image` (-DLANCZOS_WITH_STB): the same four functions then call stbi_load /
 * stbi_write_png (stb.cpp:1-6, full_TB.h:107,172).  The stb headers are third-party files of the reference tree and are not
 * copied into this repository.
 */
#ifndef LZ_IMAGE_IO_H
#define LZ_IMAGE_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Like stbi_load(path, &w, &h, &channels_in_file, desired_channels): returns malloc'd interleaved u8 HWC with
 * `desired_channels` channels (1, 3 or 4; 0 = as stored), NULL on failure. */
uint8_t* lz_image_load(const char* path, int* w, int* h, int* channels_in_file, int desired_channels);
/* Like stbi_write_png(path, w, h, comp, data, stride_bytes): 1 on success, 0 on failure. */
int lz_image_write_png(const char* path, int w, int h, int comp, const void* data, int stride_bytes);
/* Binary PPM (comp 3) / PGM (comp 1). */
int lz_image_write_pnm(const char* path, int w, int h, int comp, const void* data, int stride_bytes);
void lz_image_free(void* p);
const char* lz_image_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
