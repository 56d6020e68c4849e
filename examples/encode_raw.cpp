// encode_raw — the reference's examples/encoder_example.c on the MI355X path, through the C ABI only.
//
//   encode_raw <channels> <rate> <quality> <in.f32> <out.ogg> [data_dir]
//
// in.f32: raw interleaved little-endian float32 PCM.  One stream (S = 1) for clarity; a service would
// create the encoder for thousands of streams and hand [S][channels][1024] chunks to vbm_frontend_write.
// Build: hipcc -O2 -I../include encode_raw.cpp -L../vorbis_aotuv_lancer_amd -lvorbis_mi355x -o encode_raw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "vorbis_mi355x.h"

#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, vbm_last_error()); return 1; } } while (0)

static int drain(vbm_frontend *fe, vbm_ogg_stream *os, uint8_t *d_pkt, int *d_len, int maxb, FILE *out)
{
    vbm_packet_info info[1];
    std::vector<uint8_t> pkt(maxb);
    for (;;) {
        int n = 0, len = 0;
        CK(vbm_frontend_encode_round(fe, d_pkt, d_len, info, &n, nullptr));     // vorbis_analysis_blockout + vorbis_analysis
        if (n == 0) return 0;
        (void)hipMemcpy(&len, d_len, sizeof(int), hipMemcpyDeviceToHost);
        (void)hipMemcpy(pkt.data(), d_pkt, len, hipMemcpyDeviceToHost);
        CK(vbm_ogg_stream_packetin(os, pkt.data(), len, info[0].eos, info[0].granulepos));   // ogg_stream_packetin
        const uint8_t *page;
        long bytes;
        while (vbm_ogg_stream_pageout(os, 0, &page, &bytes) == 1) fwrite(page, 1, bytes, out);   // ogg_stream_pageout
    }
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s channels rate quality in.f32 out.ogg [data_dir]\n", argv[0]); return 2; }
    const int ch = atoi(argv[1]), rate = atoi(argv[2]);
    const std::string q = argv[3], dir = argc > 6 ? argv[6] : "../vorbis_aotuv_lancer_amd/data";
    const std::string mode = dir + "/mode_" + std::to_string(ch) + "ch_" + std::to_string(rate) + "_q" + q + ".vpk";
    FILE *in = fopen(argv[4], "rb"), *out = fopen(argv[5], "wb");
    if (!in || !out) { perror("open"); return 1; }

    vbm_setup_handle *setup;
    vbm_encoder *enc;
    vbm_frontend *fe;
    vbm_ogg_stream *os;
    CK(vbm_setup_create(&setup, (dir + "/common.vpk").c_str(), mode.c_str()));   // vorbis_encode_init_vbr
    CK(vbm_encoder_create(&enc, setup, 1, 1));                                    // vorbis_analysis_init + vorbis_block_init
    CK(vbm_frontend_create(&fe, enc));
    CK(vbm_ogg_stream_create(&os, 0x4d493335));                                   // ogg_stream_init

    // vorbis_analysis_headerout + ogg_stream_flush (encoder_example.c:139-157)
    long lens[3];
    const char *comments[] = {"ENCODER=encode_raw (MI355X)"};
    CK(vbm_header_packets(setup, nullptr, comments, 1, nullptr, 0, lens));
    std::vector<uint8_t> hdr(lens[0] + lens[1] + lens[2]);
    CK(vbm_header_packets(setup, nullptr, comments, 1, hdr.data(), (long)hdr.size(), lens));
    long at = 0;
    for (int i = 0; i < 3; at += lens[i], i++) CK(vbm_ogg_stream_packetin(os, hdr.data() + at, lens[i], 0, 0));
    const uint8_t *page;
    long bytes;
    while (vbm_ogg_stream_pageout(os, 1, &page, &bytes) == 1) fwrite(page, 1, bytes, out);

    const int maxb = vbm_encoder_max_packet_bytes(enc), CHUNK = 1024;
    float *d_pcm;
    uint8_t *d_pkt;
    int *d_len;
    if (hipMalloc((void **)&d_pcm, sizeof(float) * ch * CHUNK) != hipSuccess || hipMalloc((void **)&d_pkt, maxb) != hipSuccess ||
        hipMalloc((void **)&d_len, sizeof(int)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    std::vector<float> inter(ch * CHUNK), planar(ch * CHUNK);
    for (;;) {
        const size_t got = fread(inter.data(), sizeof(float) * ch, CHUNK, in);
        if (got == 0) break;
        for (size_t i = 0; i < got; i++)                       // uninterleave (encoder_example.c:201-206)
            for (int c = 0; c < ch; c++) planar[c * got + i] = inter[i * ch + c];
        (void)hipMemcpy(d_pcm, planar.data(), sizeof(float) * ch * got, hipMemcpyHostToDevice);
        CK(vbm_frontend_write(fe, d_pcm, (int)got, nullptr));  // vorbis_analysis_buffer + vorbis_analysis_wrote
        if (drain(fe, os, d_pkt, d_len, maxb, out)) return 1;
    }
    const int id0 = 0;
    CK(vbm_frontend_finish(fe, &id0, 1, nullptr));             // vorbis_analysis_wrote(&vd, 0)
    if (drain(fe, os, d_pkt, d_len, maxb, out)) return 1;
    while (vbm_ogg_stream_pageout(os, 1, &page, &bytes) == 1) fwrite(page, 1, bytes, out);

    fclose(out);
    fclose(in);
    vbm_ogg_stream_destroy(os);
    vbm_frontend_destroy(fe);
    vbm_encoder_destroy(enc);
    vbm_setup_destroy(setup);
    return 0;
}
