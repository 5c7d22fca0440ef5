import sys
p='/root/repo/oracle/ref_slice.c'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)
rep('''            o->partition[M] = IS_INTRA(h->mb.i_type) || h->mb.i_type == P_SKIP ? D_16x16 : h->mb.i_partition;
            for (i = 0; i < 4; i++) o->sub_partition[M * 4 + i] = h->mb.i_type == P_8x8 ? h->mb.i_sub_partition[i] : 0;''',
'''            o->partition[M] = IS_INTRA(h->mb.i_type) || IS_SKIP(h->mb.i_type) || h->mb.i_type == B_DIRECT ? D_16x16 : h->mb.i_partition;
            for (i = 0; i < 4; i++) o->sub_partition[M * 4 + i] = h->mb.i_type == P_8x8 || h->mb.i_type == B_8x8 ? h->mb.i_sub_partition[i] : 0;''')
rep('''                for (i = 0; i < 4; i++) o->ref[M * 4 + i] = h->mb.ref[0][h->mb.i_b8_xy + (i & 1) + (i >> 1) * h->mb.i_b8_stride];
''','''                for (i = 0; i < 4; i++) o->ref[M * 4 + i] = h->mb.ref[0][h->mb.i_b8_xy + (i & 1) + (i >> 1) * h->mb.i_b8_stride];
                if (o2 && o2->mv1) {
                    const int l1 = h->sh.i_type == SLICE_TYPE_B;
                    for (i = 0; i < 16; i++) {
                        int o4 = h->mb.i_b4_xy + (i & 3) + (i >> 2) * h->mb.i_b4_stride;
                        o2->mv1[(M * 16 + i) * 2] = l1 ? h->mb.mv[1][o4][0] : 0; o2->mv1[(M * 16 + i) * 2 + 1] = l1 ? h->mb.mv[1][o4][1] : 0;
                    }
                    for (i = 0; i < 4; i++) o2->ref1[M * 4 + i] = l1 ? h->mb.ref[1][h->mb.i_b8_xy + (i & 1) + (i >> 1) * h->mb.i_b8_stride] : -1;
                }
''')
rep('''            } else
                for (i = 0; i < 4; i++) o->ref[M * 4 + i] = -1;''','''            } else {
                for (i = 0; i < 4; i++) o->ref[M * 4 + i] = -1;
                if (o2 && o2->ref1) for (i = 0; i < 4; i++) o2->ref1[M * 4 + i] = -1;
            }''')
rep('''        /* x264_reference_update: newest first */
        if (n_avail == 16) x264_frame_delete(refs[--n_avail]);
        for (i = n_avail; i > 0; i--) refs[i] = refs[i - 1];
        refs[0] = h->fdec; n_avail++;
        if (n_avail > p->n_refs) x264_frame_delete(refs[--n_avail]);
        h->fdec = x264_frame_new(h);''','''        /* x264_reference_update, encoder.c:1060-1093: a disposable frame is dropped, a kept one pushes the oldest out of the DPB */
        if (!is_b) {
            if (n_avail == 16) x264_frame_delete(refs[--n_avail]);
            for (i = n_avail; i > 0; i--) refs[i] = refs[i - 1];
            refs[0] = h->fdec; n_avail++;
            if (n_avail > dpb) x264_frame_delete(refs[--n_avail]);
            h->fdec = x264_frame_new(h);
        }''')
rep('''    free(bsbuf);
    free(h);''','''    free(bsbuf); free(order); free(ftype);
    free(h);''')
open(p,'w').write(s)
print('ok')
