/* TEST INFRASTRUCTURE -- oracle/_ref/msan/refslice_msan: the reference's code and the harness around it (ref_slice.c) built with clang's
 * MemorySanitizer, as a program of its own (MSan needs an instrumented main).  It runs ONE refslice_encode_stream job read from a file
 * written by scratch/dump_ref_job.py and reports every use of an uninitialised value inside the reference -- the way the harness's
 * pinned-to-zero spots (lowres border columns, the luma corner sample) are found.  `make -C oracle msan`; CPU only. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define REFSLICE_TRACE
#include "ref_slice.c"
void __msan_set_death_callback(void (*cb)(void));
static void where(void)
{
    x264_t *h = refslice_trace_h;
    fprintf(stderr, "harness: coded frame %d (input %d, slice type %d, i_ref0 %d), macroblock %d (%d,%d)\n", refslice_trace_frame, h ? h->fenc->i_frame : -1,
            h ? h->sh.i_type : -1, h ? h->i_ref0 : -1, refslice_trace_mb, h ? h->mb.i_mb_x : -1, h ? h->mb.i_mb_y : -1);
    if (h && h->i_ref0 > 0) fprintf(stderr, "harness: fref0[0]: input %d, type %d, i_ref[0] %d, poc %d\n", h->fref0[0]->i_frame, h->fref0[0]->i_type, h->fref0[0]->i_ref[0], h->fref0[0]->i_poc);
}

int main(int argc, char **argv)
{
    refslice_params p; refslice_ext e; refslice_out o; refslice_out2 o2;
    FILE *f = fopen(argc > 1 ? argv[1] : "job.bin", "rb");
    int sp, se, i, fn = 0;
    size_t ny, nc;
    uint8_t *y, *u, *v;
    unsigned sum = 0;
    __msan_set_death_callback(where);
    if (!f) { perror("job"); return 2; }
    if (fread(&fn, 4, 1, f) != 1) return 2;                  /* 0: refslice_encode_stream (the whole encoder), 1: refslice_encode_chain2 (lock-step chains) */
    if (fread(&sp, 4, 1, f) != 1 || sp != (int)sizeof(p) || fread(&p, sizeof(p), 1, f) != 1) { fprintf(stderr, "params size %d, want %zu\n", sp, sizeof(p)); return 2; }
    if (fread(&se, 4, 1, f) != 1 || se != (int)sizeof(e) || fread(&e, sizeof(e), 1, f) != 1) { fprintf(stderr, "ext size %d, want %zu\n", se, sizeof(e)); return 2; }
    ny = (size_t)p.width * p.height * p.n_frames; nc = ny / 4;
    y = malloc(ny); u = malloc(nc); v = malloc(nc);
    if (fread(y, 1, ny, f) != ny || fread(u, 1, nc, f) != nc || fread(v, 1, nc, f) != nc) { fprintf(stderr, "short clip\n"); return 2; }
    fclose(f);
    for (i = 0; i < (int)(sizeof(o) / sizeof(void *)); i++) ((void **)&o)[i] = calloc(1, 16 << 20);
    for (i = 0; i < (int)(sizeof(o2) / sizeof(void *)); i++) ((void **)&o2)[i] = calloc(1, 16 << 20);
    i = fn ? refslice_encode_chain2(&p, &e, y, u, v, &o, &o2) : refslice_encode_stream(&p, &e, y, u, v, &o, &o2);
    printf("rc %d\n", i);
    for (i = 0; i < p.n_frames; i++) {
        int k, n = o2.payload_len[i];
        for (k = 0; k < n; k++) sum = sum * 31 + o2.payload[(size_t)i * e.payload_cap + k];
        printf("frame %d: %d bytes\n", i, n);
    }
    printf("checksum %08x\n", sum);
    return 0;
}
