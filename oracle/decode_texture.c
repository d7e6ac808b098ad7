/* decode_texture.c -- TEST/ASSET INFRASTRUCTURE ONLY (build container only).
 *
 * The reference decodes its JPEG textures with the stb_image.h it vendors (src/image_io.h:26:
 * stbi_load(path, &w, &h, &n, 3)).  JPEG decoders differ by a level or two per texel (IDCT and chroma
 * upsampling), so to compare pixels with the reference's own output images the texels must come from the same
 * decoder.  This tool compiles that header from where it lies under /root/reference (oracle/Makefile,
 * target _ref/decode_texture; nothing is copied into the repository) and writes the decoded RGB8 pixels as a
 * binary PPM -- data, which is what assets/ holds.
 *
 * usage: decode_texture in.jpg out.ppm */
#include <stdio.h>
#define STB_IMAGE_IMPLEMENTATION
#include STB_IMAGE_HEADER

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s in.jpg out.ppm\n", argv[0]); return 2; }
    int w = 0, h = 0, n = 0;
    unsigned char* px = stbi_load(argv[1], &w, &h, &n, 3);
    if (!px) { fprintf(stderr, "stbi_load failed for '%s': %s\n", argv[1], stbi_failure_reason()); return 1; }
    FILE* f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    fwrite(px, 1, (size_t)w * h * 3, f);
    fclose(f);
    stbi_image_free(px);
    printf("%s: %dx%d (%d channels in file)\n", argv[2], w, h, n);
    return 0;
}
