/* encoder_compat — a plain-C encoder written against the reference's own API names, linked with
 * libvorbis_mi355x.so instead of libvorbis/libvorbisenc/libogg.  The call sequence is the one of the
 * reference's examples/encoder_example.c:127-244 (init_vbr, comment, analysis_init, block_init, headerout,
 * then buffer / wrote / blockout / analysis / bitrate_addblock / bitrate_flushpacket / page out), so the
 * same program text builds against <vorbis/codec.h> + <vorbis/vorbisenc.h> + <ogg/ogg.h> as well: only
 * the #include line differs.
 *
 *   encoder_compat <channels> <rate> <quality> [--f32] [--no-eos] [--dump packets.pkt] < pcm > out.ogg
 *
 * stdin: raw interleaved PCM, signed 16-bit little endian (what encoder_example.c reads after its WAV
 * header) or, with --f32, 32-bit floats.  --dump writes every audio packet as <int32 length><bytes>
 * (the layout of the reference-build dumps under tests/golden/).  --no-eos stops after the last chunk
 * without vorbis_analysis_wrote(vd, 0), as the survey's probe driver did.
 * Build: gcc -O2 -I../include encoder_compat.c -L../vorbis_aotuv_lancer_amd -lvorbis_mi355x
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vorbis_compat.h" /* reference: <vorbis/vorbisenc.h> (which pulls codec.h and ogg/ogg.h) */

#define READ 1024

static void put_page(const ogg_page *og)
{
    fwrite(og->header, 1, og->header_len, stdout);
    fwrite(og->body, 1, og->body_len, stdout);
}

int main(int argc, char **argv)
{
    ogg_stream_state os; /* packets -> pages */
    ogg_page og;
    ogg_packet op;
    vorbis_info vi;
    vorbis_comment vc;
    vorbis_dsp_state vd;
    vorbis_block vb;
    int eos = 0, f32 = 0, no_eos = 0, i, ret;
    FILE *dump = NULL;
    long packets = 0;

    if (argc < 4) {
        fprintf(stderr, "usage: %s channels rate quality [--f32] [--no-eos] [--dump file] < pcm > out.ogg\n", argv[0]);
        return 2;
    }
    const int ch = atoi(argv[1]);
    const long rate = atol(argv[2]);
    const float quality = (float)atof(argv[3]);
    for (i = 4; i < argc; i++) {
        if (!strcmp(argv[i], "--f32")) f32 = 1;
        else if (!strcmp(argv[i], "--no-eos")) no_eos = 1;
        else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = fopen(argv[++i], "wb");
    }
    const size_t frame = (size_t)ch * (f32 ? 4 : 2);
    unsigned char *readbuffer = (unsigned char *)malloc(READ * frame);

    vorbis_info_init(&vi);
    ret = vorbis_encode_init_vbr(&vi, ch, rate, quality);
    if (ret) {
        fprintf(stderr, "vorbis_encode_init_vbr: %d (no mode pack for this channels/rate/quality?)\n", ret);
        return 1;
    }
    vorbis_comment_init(&vc);
    vorbis_comment_add_tag(&vc, "ENCODER", "encoder_compat.c");
    if (vorbis_analysis_init(&vd, &vi)) {
        fprintf(stderr, "vorbis_analysis_init failed (no HIP device?)\n");
        return 1;
    }
    vorbis_block_init(&vd, &vb);
    ogg_stream_init(&os, 0x4d493335);

    {   /* the three header packets, then a page break so that audio starts on a fresh page */
        ogg_packet header, header_comm, header_code;
        vorbis_analysis_headerout(&vd, &vc, &header, &header_comm, &header_code);
        ogg_stream_packetin(&os, &header);
        ogg_stream_packetin(&os, &header_comm);
        ogg_stream_packetin(&os, &header_code);
        while (ogg_stream_flush(&os, &og)) put_page(&og);
    }

    while (!eos) {
        const size_t got = fread(readbuffer, frame, READ, stdin);
        if (got == 0) {
            if (no_eos) break;
            vorbis_analysis_wrote(&vd, 0); /* end of stream: the library pads and marks the last packet */
        } else {
            float **buffer = vorbis_analysis_buffer(&vd, READ);
            size_t k;
            int c;
            for (k = 0; k < got; k++)
                for (c = 0; c < ch; c++) {
                    if (f32) {
                        float x;
                        memcpy(&x, readbuffer + (k * ch + c) * 4, 4);
                        buffer[c][k] = x;
                    } else {
                        const unsigned char *p = readbuffer + (k * ch + c) * 2;
                        buffer[c][k] = (int16_t)(p[0] | (p[1] << 8)) / 32768.f;
                    }
                }
            vorbis_analysis_wrote(&vd, (int)got);
        }
        while (vorbis_analysis_blockout(&vd, &vb) == 1) {
            vorbis_analysis(&vb, NULL);
            vorbis_bitrate_addblock(&vb);
            while (vorbis_bitrate_flushpacket(&vd, &op)) {
                if (dump) {
                    const int32_t n = (int32_t)op.bytes;
                    fwrite(&n, 4, 1, dump);
                    fwrite(op.packet, 1, op.bytes, dump);
                }
                packets++;
                ogg_stream_packetin(&os, &op);
                while (!eos) {
                    if (!ogg_stream_pageout(&os, &og)) break;
                    put_page(&og);
                    if (ogg_page_eos(&og)) eos = 1;
                }
            }
        }
    }
    if (no_eos)
        while (ogg_stream_flush(&os, &og)) put_page(&og);

    ogg_stream_clear(&os);
    vorbis_block_clear(&vb);
    vorbis_dsp_clear(&vd);
    vorbis_comment_clear(&vc);
    vorbis_info_clear(&vi);
    if (dump) fclose(dump);
    free(readbuffer);
    fprintf(stderr, "encoder_compat: %ld audio packets\n", packets);
    return 0;
}
