/* oracle/ref_genmusic_wrap.c -- TEST INFRASTRUCTURE ONLY.
 * Makes the reference's own generate_music() (opus-fix/tests/test_opus_encode.c:59-88) callable: the reference's test
 * source is compiled IN PLACE (included by path, nothing copied; its main() renamed on the command line) and one entry
 * point seeds its file-static generator and calls it. Built into oracle/_ref/librefgen.so by oracle/Makefile. */
#include REF_TEST_OPUS_ENCODE_C

void refgen_music(short *buf, int len, unsigned seed, int skip)
{
    Rz = Rw = seed;
    while (skip-- > 0) (void)fast_rand();
    generate_music(buf, len);
}
