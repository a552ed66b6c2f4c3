/*
 * image_io.h -- interleaved 8-bit image load/store for the harness.
 *
 * The reference's harness uses stb_image / stb_image_write (stb.cpp, full_TB.h:107,172): stbi_load returns
 * interleaved u8 HWC with the requested channel count; stbi_write_png takes the same layout.  This is our own
 * small codec with the SAME buffer contract (so a caller that already links stb can keep using it and just
 * hand the buffers to lanczos_u8): PNG (8-bit gray / gray+alpha / RGB / RGBA, non-interlaced) through zlib,
 * and binary PPM/PGM.
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
