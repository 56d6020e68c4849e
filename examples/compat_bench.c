/* compat_bench — throughput of the DROP-IN path: many encoder streams driven through the reference's own entry
 * points (include/vorbis_compat.h), host PCM in, packets back on the host; the loop per stream is the one of the
 * reference's examples/encoder_example.c:179-236:
 *     vorbis_analysis_buffer / vorbis_analysis_wrote
 *     while (vorbis_analysis_blockout(&vd, &vb) == 1) {
 *         vorbis_analysis(&vb, NULL); vorbis_bitrate_addblock(&vb);
 *         while (vorbis_bitrate_flushpacket(&vd, &op)) consume(op);
 *     }
 * T host threads own M streams each; every thread writes one READ-sample chunk to each of its streams, then drains
 * each of them.  Streams share device pools of `pool` slots (VORBIS_MI355X_POOL_STREAMS): one device round carves and
 * encodes a block for every stream of a pool.  Plain C (gcc, pthreads), no HIP headers.
 *
 *   compat_bench <threads> <streams_per_thread> <pool_streams> <writes> [warmup_writes]
 * prints one JSON line: audio seconds encoded per wall second (= streams at 1x realtime), device rounds, packet bytes.
 */
#define _DEFAULT_SOURCE
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "vorbis_compat.h"

#define READ 1024
#define CH 2
#define RATE 44100

typedef struct {
    int id, nstreams, writes, warmup;
    vorbis_dsp_state *vd;      /* this thread's streams: made by main() one thread after the other, so that they fill whole pools */
    vorbis_block *vb;
    vorbis_info *vi;
    pthread_barrier_t *bar;
    const float *pcm;          /* [CH][period] planar source, shared */
    long period;
    long long bytes, packets, samples;
    double t0, t1, t_write, t_drain;
    int failed;
} worker;

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void *run(void *arg)
{
    worker *w = (worker *)arg;
    const int M = w->nstreams;
    vorbis_dsp_state *vd = w->vd;
    vorbis_block *vb = w->vb;
    ogg_packet op;
    pthread_barrier_wait(w->bar);
    for (int k = 0; k < w->warmup + w->writes && !w->failed; k++) {
        if (k == w->warmup) {
            pthread_barrier_wait(w->bar);
            w->t0 = now();
            w->bytes = w->packets = w->samples = 0;
            w->t_write = w->t_drain = 0;
        }
        const double ta = now();
        for (int s = 0; s < M; s++) {
            float **buf = vorbis_analysis_buffer(&vd[s], READ);
            /* every stream reads the shared signal at its own offset (streams differ, nothing is generated in the loop) */
            const long at = ((long)(w->id * M + s) * 7919 + (long)k * READ) % w->period;
            const long first = (w->period - at < READ) ? w->period - at : READ;
            for (int c = 0; c < CH; c++) {
                memcpy(buf[c], w->pcm + c * w->period + at, first * sizeof(float));
                if (first < READ) memcpy(buf[c] + first, w->pcm + c * w->period, (READ - first) * sizeof(float));
            }
            if (vorbis_analysis_wrote(&vd[s], READ)) { w->failed = 1; break; }
        }
        const double tb = now();
        w->t_write += tb - ta;
        for (int s = 0; s < M && !w->failed; s++) {
            while (vorbis_analysis_blockout(&vd[s], &vb[s]) == 1) {
                if (vorbis_analysis(&vb[s], NULL) || vorbis_bitrate_addblock(&vb[s])) { w->failed = 1; break; }
                while (vorbis_bitrate_flushpacket(&vd[s], &op)) {
                    w->bytes += op.bytes;
                    w->packets++;
                    /* what the packet advances the stream by: the distance between block centres (lib/block.c:745-759) */
                    w->samples += (vb[s].W ? 2048 : 256) / 4 + (vb[s].nW ? 2048 : 256) / 4;
                }
            }
        }
        w->t_drain += now() - tb;
    }
    w->t1 = now();
    pthread_barrier_wait(w->bar);
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s threads streams_per_thread pool_streams writes [warmup]\n", argv[0]);
        return 2;
    }
    const int T = atoi(argv[1]), M = atoi(argv[2]);
    int pool = atoi(argv[3]);
    const int writes = atoi(argv[4]), warmup = argc > 5 ? atoi(argv[5]) : 8;
    if (T < 1 || M < 1 || pool < 1 || writes < 1) return 2;
    if (vorbis_mi355x_ctl(VORBIS_MI355X_POOL_STREAMS, &pool)) return 3;

    /* SURVEY 8(d)-shaped signal: two sines + noise + a burst every 1.33 s, 4 s long, shared by all streams */
    const long period = 4L * RATE;
    float *pcm = (float *)malloc(sizeof(float) * period * CH);
    uint32_t lcg = 12345u;
    for (long i = 0; i < period; i++)
        for (int c = 0; c < CH; c++) {
            lcg = lcg * 1664525u + 1013904223u;
            const double noise = ((lcg >> 8) / 8388608.0) - 1.0;
            const double t = (double)i / RATE;
            double x = 0.3 * sin(2 * M_PI * 440.0 * (c + 1) * t) + 0.2 * sin(2 * M_PI * 3000.0 * t + c) + 0.05 * noise;
            if ((i % (RATE * 4 / 3)) < 200) x += 0.6 * noise;
            pcm[c * period + i] = (float)x;
        }

    vorbis_info vi;
    vorbis_info_init(&vi);
    if (vorbis_encode_init_vbr(&vi, CH, RATE, 0.5f)) { fprintf(stderr, "no q5 stereo mode pack\n"); return 4; }
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, T);
    worker *w = (worker *)calloc(T, sizeof(*w));
    pthread_t *th = (pthread_t *)calloc(T, sizeof(*th));
    /* slots are handed out pool after pool in the order streams are made: thread i's M streams first, then thread i + 1's —
     * with pool_streams == M (or a divisor) a pool belongs to one thread, whose rounds then never wait for another thread's */
    for (int i = 0; i < T; i++) {
        w[i].id = i; w[i].nstreams = M; w[i].writes = writes; w[i].warmup = warmup;
        w[i].vi = &vi; w[i].bar = &bar; w[i].pcm = pcm; w[i].period = period;
        w[i].vd = (vorbis_dsp_state *)calloc(M, sizeof(vorbis_dsp_state));
        w[i].vb = (vorbis_block *)calloc(M, sizeof(vorbis_block));
        for (int s = 0; s < M; s++)
            if (vorbis_analysis_init(&w[i].vd[s], &vi) || vorbis_block_init(&w[i].vd[s], &w[i].vb[s])) { fprintf(stderr, "vorbis_analysis_init failed\n"); return 5; }
    }
    for (int i = 0; i < T; i++) pthread_create(&th[i], NULL, run, &w[i]);
    double t0 = 1e300, t1 = 0;
    long long bytes = 0, packets = 0, samples = 0;
    int failed = 0;
    for (int i = 0; i < T; i++) {
        pthread_join(th[i], NULL);
        if (w[i].t0 < t0) t0 = w[i].t0;
        if (w[i].t1 > t1) t1 = w[i].t1;
        bytes += w[i].bytes; packets += w[i].packets; samples += w[i].samples;
        failed |= w[i].failed;
    }
    long long rounds = 0;
    vorbis_mi355x_ctl(VORBIS_MI355X_ROUNDS, &rounds);
    double lib[8], tw = 0, td = 0;
    vorbis_mi355x_ctl(VORBIS_MI355X_TIMES, lib);
    for (int i = 0; i < T; i++) { tw += w[i].t_write; td += w[i].t_drain; }
    fprintf(stderr, "host seconds (all threads, timed region + warm-up for the library's): app write phase %.3f, drain phase %.3f; "
            "library: wrote-copy %.3f, staging copies %.3f, upload %.3f, rounds %.3f, compaction + D2H %.3f, filing %.3f\n",
            tw, td, lib[0], lib[1], lib[2], lib[3], lib[4], lib[5]);
    const double wall = t1 - t0, audio = (double)samples / RATE;
    printf("{\"threads\": %d, \"streams\": %d, \"pool_streams\": %d, \"writes\": %d, \"wall_s\": %.4f, \"value\": %.1f, "
           "\"input_audio_s\": %.1f, \"encoded_audio_s\": %.1f, \"packets\": %lld, \"packet_bytes\": %lld, "
           "\"device_rounds_total\": %lld, \"failed\": %d}\n",
           T, T * M, pool, writes, wall, audio / wall, (double)T * M * writes * READ / RATE, audio, packets, bytes, rounds, failed);
    for (int i = 0; i < T; i++)
        for (int s = 0; s < M; s++) {
            vorbis_block_clear(&w[i].vb[s]);
            vorbis_dsp_clear(&w[i].vd[s]);
        }
    vorbis_info_clear(&vi);
    free(pcm);
    return failed ? 1 : 0;
}
