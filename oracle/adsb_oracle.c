/*
 * TEST INFRASTRUCTURE ONLY -- see adsb_oracle.h.
 *
 * CPU restatement of the reference ADS-B chain as functions of a whole
 * sample array.  Build with -ffp-contract=off: the reference is compiled
 * as ISO C99 (configure:58), which forbids contraction.
 */
#include "adsb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ a1 -- */

size_t
oracle_magnitude(const void *iq, size_t nbytes, int encoding, float *m) {
    size_t n, i;

    switch (encoding) {
    case ORACLE_ENC_UC8: {
	/* plane-iq.c:131-136: ((b - 128) / 128.f)^2 summed, in float */
	const unsigned char *b = iq;
	n = nbytes / 2;
	for (i = 0; i < n; i++) {
	    float re = (b[2 * i] - 128) / 128.f;
	    float im = (b[2 * i + 1] - 128) / 128.f;
	    m[i] = re * re + im * im;
	}
	return n;
    }
    case ORACLE_ENC_SC8: {
	/* same formula with radio-file.c:137 scaling (b / 128) */
	const signed char *b = iq;
	n = nbytes / 2;
	for (i = 0; i < n; i++) {
	    float re = b[2 * i] / 128.f;
	    float im = b[2 * i + 1] / 128.f;
	    m[i] = re * re + im * im;
	}
	return n;
    }
    case ORACLE_ENC_SC16: {
	/*
	 * radio-file.c:140 scales sc16 by 1/32768; the reference never routes
	 * it into ADS-B, so the magnitude is a definition of this build
	 * (SURVEY.md section 8a, sc16 note; parity unpinned): the exact integer
	 * I^2 + Q^2 (< 2^32), ONE correctly rounded conversion to float, and an
	 * exact power-of-two scale.  It differs from squaring and adding in
	 * float by at most 1 ulp and does not depend on contraction.
	 */
	const unsigned char *b = iq;
	n = nbytes / 4;
	for (i = 0; i < n; i++) {
	    int32_t vi = (int16_t) (b[4 * i] | (b[4 * i + 1] << 8));
	    int32_t vq = (int16_t) (b[4 * i + 2] | (b[4 * i + 3] << 8));
	    uint32_t sum = (uint32_t) (vi * vi) + (uint32_t) (vq * vq);
	    m[i] = (float) sum * (1.0f / 1073741824.0f);
	}
	return n;
    }
    default:
	return 0;
    }
}

/* ------------------------------------------------------------------ a3 -- */

int
oracle_preamble(const float *m, size_t p) {
    /* plane-iq.c:173-175, tmp[i] == m[p + i]; all comparisons strict */
    const float *t = m + p;

    return t[1] < t[0] && t[2] > t[1] && t[3] < t[2] &&
	    t[4] < t[0] && t[5] < t[0] && t[6] < t[0] &&
	    t[8] < t[7] && t[9] > t[8];
}

/* --------------------------------------------------------------- a5, a6 -- */

void
oracle_slice(const float *m, size_t p, char *bits, double *amplitude_56) {
    /*
     * plane-iq.c:58-86.  A message created at iteration p is fed on
     * iterations p, p+2, ..., p+222 (parity toggles, :67-69) with
     * s0 = signal[q+16], s1 = signal[q+17] (:179-181).
     */
    double a56 = 0;
    int k;

    for (k = 0; k < ORACLE_MSG_BITS; k++) {
	float s0 = m[p + 16 + 2 * k];
	float s1 = m[p + 17 + 2 * k];
	double amplitude = (s0 + s1) / 2;	/* float add, :62 */

	if (k < 56)
	    a56 += amplitude;			/* :72-73 */
	bits[k] = '0' + (s1 <= s0);		/* :65, :75: a tie is a 1 */
    }
    bits[ORACLE_MSG_BITS] = '\0';
    *amplitude_56 = a56;
}

/* ------------------------------------------------------------------ a7 -- */

static uint64_t
bits_field(const char *s, int off, int n) {
    /* plane.c:351-361, MSB first */
    uint64_t v = 0;
    int i;

    for (i = 0; i < n; i++)
	v = (v << 1) | (uint64_t) (s[off + i] == '1');
    return v;
}

/* ------------------------------------------------------------------ a8 -- */

uint32_t
oracle_crc24(const char *bits, int size) {
    /* plane.c:373-385: generator 0x1FFF409, init 0, MSB first */
    uint32_t r = 0;
    int i;

    for (i = 0; i < size; i++) {
	r ^= (uint32_t) (bits[i] == '1') << 23;
	r <<= 1;
	if (r & 0x1000000u)
	    r ^= 0x1fff409u;
    }
    return r;
}

/* --------------------------------------------------- plane set, plane.c -- */

struct oracle_plane {			/* plane.c:17-53 */
    uint32_t address;
    int used;
    int have_flight_id, have_squawk, have_altitude, have_velocity, have_coords;
    int velocity_we, velocity_sn;
    long altitude;
    unsigned int enc_lat[2], enc_lon[2];
    long long enc_sec[2], enc_usec[2];
    double latitude, longitude;
    long long update_sec, update_usec;
    char flight_id[9];
    char squawk_id[5];
    double last_amplitude;
};

struct oracle_plane_set {
    struct oracle_plane *slots;
    size_t nslots;			/* power of two */
    size_t count;
    struct oracle_plane *last;
};

struct oracle_plane_set *
oracle_plane_set_new(void) {
    struct oracle_plane_set *ps = calloc(1, sizeof *ps);

    ps->nslots = 1024;
    ps->slots = calloc(ps->nslots, sizeof *ps->slots);
    return ps;
}

void
oracle_plane_set_delete(struct oracle_plane_set *ps) {
    if (ps == NULL)
	return;
    free(ps->slots);
    free(ps);
}

size_t
oracle_plane_set_count(const struct oracle_plane_set *ps) {
    return ps->count;
}

static struct oracle_plane *
pset_find(struct oracle_plane_set *ps, uint32_t address, int create) {
    size_t mask, i;

    if (create && (ps->count + 1) * 2 > ps->nslots) {
	struct oracle_plane *old = ps->slots;
	size_t oldn = ps->nslots, j;
	long last_idx = ps->last != NULL? (long) (ps->last - old) : -1;
	uint32_t last_addr = last_idx >= 0? old[last_idx].address : 0;

	ps->nslots *= 2;
	ps->slots = calloc(ps->nslots, sizeof *ps->slots);
	ps->count = 0;
	for (j = 0; j < oldn; j++)
	    if (old[j].used) {
		struct oracle_plane *np = pset_find(ps, old[j].address, 1);
		*np = old[j];
	    }
	free(old);
	ps->last = last_idx >= 0? pset_find(ps, last_addr, 0) : NULL;
    }

    mask = ps->nslots - 1;
    i = (address * 2654435761u) & mask;
    for (;; i = (i + 1) & mask) {
	struct oracle_plane *pl = &ps->slots[i];
	if (!pl->used) {
	    if (!create)
		return NULL;
	    memset(pl, 0, sizeof *pl);	/* plane.c:116 */
	    pl->used = 1;
	    pl->address = address;
	    ps->count++;
	    return pl;
	}
	if (pl->address == address)
	    return pl;
    }
}

/* ----------------------------------------------------------------- a11 -- */

static void
decode_cpr(struct oracle_plane *pl) {
    /* plane.c:387-450 */
    long long diff = pl->enc_sec[1] - pl->enc_sec[0];
    int odd, j, rlat, arlat, nl;
    double lat, lon;

    pl->have_coords = 0;
    if (llabs(diff) > 10)
	return;
    odd = diff > 0;

    j = floor((59 * (double) pl->enc_lat[0] - 60 * (double) pl->enc_lat[1]) /
	    (1 << 17) + 0.5);
    if (j < 0)
	j += 60 - odd;
    if (!((j >= 0 && j <= 15) || (j >= 45 && j <= 60)))
	return;
    if (j >= 45)
	j -= 60;

    lat = (j + (double) pl->enc_lat[odd] / (1 << 17)) * 360 / (60 - odd);
    pl->latitude = lat;

    rlat = (lat < 0? ceil(lat / 6) : floor(lat / 6)) * 6 + 3;
    arlat = abs(rlat);
    if (arlat <= 3)
	nl = 59;
    else if (arlat == 87)
	nl = 2;
    else if (arlat > 87)
	nl = 1;
    else {
	double c = cos(M_PI * arlat / 180);
	nl = floor(2 * M_PI / acos(1 - (1 - cos(M_PI / 30)) / (c * c)));
    }

    j = floor(((nl - 1) * (double) pl->enc_lon[0] -
	    nl * (double) pl->enc_lon[1]) / (1 << 17) + 0.5);
    if (j < 0)
	j += nl - odd;

    lon = (j + (double) pl->enc_lon[odd] / (1 << 17)) * 360.0 / (nl - odd);
    if (lon >= 180)
	lon -= 360;
    pl->longitude = lon;
    pl->have_coords = 1;
}

/* ------------------------------------------------------------- a9, a10 -- */

static char
callsign_char(unsigned v) {		/* plane.c:337-349 */
    if (v >= 1 && v <= 26)
	return 'A' + v - 1;
    if (v == 32)
	return ' ';
    if (v >= 48 && v <= 57)
	return '0' + v - 48;
    return '?';
}

#define XBIT(x, o) (((x) >> (o)) & 1u)

static unsigned
ac_digit(unsigned x, int n) {		/* plane.c:369-371 */
    return XBIT(x, n) | (XBIT(x, n + 2) << 1) | (XBIT(x, n + 4) << 2);
}

static unsigned
id_digit(unsigned x, int n) {		/* plane.c:366-368 */
    return (XBIT(x, n) << 2) | (XBIT(x, n + 2) << 1) | XBIT(x, n + 4);
}

int
oracle_parse_message(struct oracle_plane_set *ps, const char *bits,
	long long tv_sec, long long tv_usec, double amplitude,
	struct oracle_frame *info) {
    static const int c_digit[2][8] = {	/* plane.c:452-455 */
	{ -1, 0, 2, 1, 4, -1, 3, -1 },
	{ -1, 4, 2, 3, 0, -1, 1, -1 }
    };
    int df = (int) bits_field(bits, 0, 5);
    int is_long = df >= 16 && df <= 24;			/* plane.c:462 */
    uint32_t ap = (uint32_t) bits_field(bits, is_long? 88 : 32, 24);
    unsigned ac = (unsigned) bits_field(bits, 19, 13);
    uint32_t crc = oracle_crc24(bits, is_long? 88 : 32);
    uint32_t address;
    struct oracle_plane *pl;

    if (df == 11 || df == 17 || df == 18) {		/* plane.c:473-478 */
	address = (uint32_t) bits_field(bits, 8, 24);
	crc ^= ap;
	if (info != NULL) {
	    info->df = df;
	    info->long_message = is_long;
	    info->syndrome = crc;
	    info->address = address;
	}
	if (crc != 0)
	    return 0;
	pl = pset_find(ps, address, 1);
    } else {						/* plane.c:479-484 */
	address = crc ^ ap;
	if (info != NULL) {
	    info->df = df;
	    info->long_message = is_long;
	    info->syndrome = address;
	    info->address = address;
	}
	pl = pset_find(ps, address, 0);
	if (pl == NULL)
	    return 0;
    }
    ps->last = pl;

    pl->update_sec = tv_sec;				/* plane.c:486-487 */
    pl->update_usec = tv_usec;
    pl->last_amplitude = amplitude;

    if (df == 17) {
	unsigned type = (unsigned) bits_field(bits, 32, 5);

	if (type >= 1 && type <= 4) {			/* plane.c:493-502 */
	    int i;
	    for (i = 0; i < 8; i++)
		pl->flight_id[i] =
			callsign_char((unsigned) bits_field(bits, 40 + 6 * i, 6));
	    pl->flight_id[8] = '\0';
	    pl->have_flight_id = 1;
	}
	if (type >= 9 && type <= 18) {			/* plane.c:503-519 */
	    unsigned alt = (unsigned) bits_field(bits, 40, 12);
	    int odd;

	    if (XBIT(alt, 4)) {
		int v = (int) ((((alt >> 5) & 0x7f) << 4) | (alt & 0xf));
		pl->altitude = (long) v * 25 - 1000;
		pl->have_altitude = 1;
	    }
	    odd = bits_field(bits, 53, 1) == 1;
	    pl->enc_lat[odd] = (unsigned) bits_field(bits, 54, 17);
	    pl->enc_lon[odd] = (unsigned) bits_field(bits, 71, 17);
	    pl->enc_sec[odd] = tv_sec;
	    pl->enc_usec[odd] = tv_usec;
	    decode_cpr(pl);
	} else if (type == 19) {			/* plane.c:522-550 */
	    unsigned subtype = (unsigned) bits_field(bits, 37, 3);

	    if (subtype == 1 || subtype == 2) {
		double we = (double) bits_field(bits, 46, 10);
		double sn = (double) bits_field(bits, 57, 10);

		/* the reference tests w_e first and leaves on zero before
		 * it ever reads s_n; neither path has a side effect */
		if (we != 0 && sn != 0) {
		    we -= 1;
		    if (bits_field(bits, 45, 1) == 1)
			we = -we;
		    sn -= 1;
		    if (bits_field(bits, 56, 1) == 1)
			sn = -sn;
		    if (subtype == 2) {
			we *= 4;
			sn *= 4;
		    }
		    pl->velocity_we = (int) we;
		    pl->velocity_sn = (int) sn;
		    pl->have_velocity = 1;
		}
	    }
	}
    } else if (df == 4) {				/* plane.c:559-585 */
	if (XBIT(ac, 6) == 0) {
	    if (XBIT(ac, 4) == 1) {
		int v = (int) ((((ac >> 7) & 0x3f) << 5) | (XBIT(ac, 5) << 4) |
			(ac & 0xf));
		pl->altitude = (long) v * 25 - 1000;
		pl->have_altitude = 1;
	    } else {
		unsigned dab = ((ac_digit(ac, 0) & 3) << 6) |
			(ac_digit(ac, 7) << 3) | ac_digit(ac, 1);
		uint16_t g = (uint16_t) dab;		/* plane.c:327-335 */
		int code, adj;

		g ^= g >> 8;
		g ^= g >> 4;
		g ^= g >> 2;
		g ^= g >> 1;
		code = g;
		adj = c_digit[code & 1][ac_digit(ac, 8)];
		if (adj != -1) {
		    pl->altitude = (long) (code * 5 - 12 + adj) * 100;
		    pl->have_altitude = 1;
		}
	    }
	}
    } else if (df == 5) {				/* plane.c:586-593 */
	pl->squawk_id[0] = (char) (id_digit(ac, 7) + '0');
	pl->squawk_id[1] = (char) (id_digit(ac, 1) + '0');
	pl->squawk_id[2] = (char) (id_digit(ac, 8) + '0');
	pl->squawk_id[3] = (char) (id_digit(ac, 0) + '0');
	pl->squawk_id[4] = '\0';
	pl->have_squawk = 1;
    }
    /* DF 11, 18, 20, 21 and everything else: state touch only */

    return 1;
}

/* ----------------------------------------------------------------- a13 -- */

int
oracle_print_last(struct oracle_plane_set *ps, char *buf, size_t len) {
    /* plane.c:598-632 */
    const struct oracle_plane *p = ps->last;
    size_t o = 0;
    time_t t;
    struct tm *tm;

    if (p == NULL || len == 0)
	return 0;
#define EMIT(...) do { \
	int r_ = snprintf(buf + o, o < len? len - o : 0, __VA_ARGS__); \
	if (r_ > 0) o += (size_t) r_; \
    } while (0)
    EMIT("%06X ", (unsigned) p->address);
    if (p->have_flight_id)
	EMIT("%-8s ", p->flight_id);
    else
	EMIT("         ");
    if (p->have_squawk)
	EMIT("%4s ", p->squawk_id);
    else
	EMIT("     ");
    if (p->have_altitude)
	EMIT("%5ldft ", p->altitude);
    else
	EMIT("        ");
    if (p->have_velocity) {
	double we = p->velocity_we;
	double sn = p->velocity_sn;
	double heading = atan2(we, sn) * 180 / M_PI;
	if (heading < 0)
	    heading += 360;
	EMIT("%6.1fkt %05.1fd ", sqrt(we * we + sn * sn), heading);
    } else
	EMIT("                ");
    if (p->have_coords)
	EMIT("% 8.5f % 8.5f ", p->latitude, p->longitude);
    else
	EMIT("                   ");
    t = (time_t) p->update_sec;
    tm = localtime(&t);
    EMIT("%02d:%02d:%02d", tm->tm_hour, tm->tm_min, tm->tm_sec);
    EMIT(" %5.1f", p->last_amplitude);
#undef EMIT
    return (int) o;
}

/* ------------------------------------------------------ the whole chain -- */

static void
stream_time(uint64_t position, long long t0, unsigned long long rate,
	long long *sec, long long *usec) {
    /* same clock as oracle/ref_shim_plane_iq.c */
    *sec = t0 + (long long) (position / rate);
    *usec = (long long) ((position % rate) * 1000000ull / rate);
}

size_t
oracle_count_candidates(const float *m, size_t n) {
    size_t p, c = 0;

    if (n < ORACLE_WINDOW)
	return 0;
    for (p = 0; p + ORACLE_WINDOW <= n; p++)
	c += (size_t) oracle_preamble(m, p);
    return c;
}

/* ------------------------------------------- D3: single-bit correction -- */

static int g_correction = 0;

void
oracle_set_correction(int on) {
    g_correction = on != 0;
}

int
oracle_correctable_bit(uint32_t syndrome, int nbits) {
    /*
     * Flipping bit b (0 = first) of an nbits-long message adds x^(nbits-1-b)
     * mod G to crc ^ ap, for data and parity bits alike.  The five DF bits
     * are never "corrected": that would change the message length and type.
     */
    uint32_t v = 1;
    int k;

    if (syndrome == 0)
	return -1;
    for (k = 0; k <= nbits - 6; k++) {
	if (v == syndrome)
	    return nbits - 1 - k;
	v <<= 1;
	if (v & 0x1000000u)
	    v ^= 0x1fff409u;
    }
    return -1;
}

/* parse; with correction on, retry a squitter once with the indicated bit
 * flipped.  bits is modified in place when a correction is made. */
static int
parse_with_correction(struct oracle_plane_set *ps, char *bits, long long sec,
	long long usec, double amplitude, struct oracle_frame *f) {
    int accepted = oracle_parse_message(ps, bits, sec, usec, amplitude, f);

    f->corrected_bit = -1;
    if (!accepted && g_correction &&
	    (f->df == 11 || f->df == 17 || f->df == 18) && f->syndrome != 0) {
	int b = oracle_correctable_bit(f->syndrome, f->long_message? 112 : 56);
	if (b >= 0) {
	    uint32_t before = f->syndrome;
	    bits[b] ^= '0' ^ '1';
	    accepted = oracle_parse_message(ps, bits, sec, usec, amplitude, f);
	    f->syndrome = before;
	    f->corrected_bit = b;
	}
    }
    return accepted;
}

size_t
oracle_scan(const float *m, size_t n, uint64_t base_position,
	struct oracle_plane_set *ps, int accepted_only,
	struct oracle_frame *out, size_t cap) {
    size_t p, count = 0;

    /*
     * plane-iq.c:148-166 processes iteration q while 18 samples remain, so a
     * message started at p completes (iteration p+222) iff p + 240 <= n.
     * Messages complete one per iteration at most, in increasing p.
     */
    if (n < ORACLE_WINDOW)
	return 0;
    for (p = 0; p + ORACLE_WINDOW <= n; p++) {
	struct oracle_frame f;

	if (!oracle_preamble(m, p))
	    continue;
	memset(&f, 0, sizeof f);
	f.position = base_position + p;
	oracle_slice(m, p, f.bits, &f.amplitude_56);
	f.amplitude = f.amplitude_56 / 56;		/* plane-iq.c:189 */
	{
	    long long sec, usec;
	    stream_time(f.position, 1700000000ll, 2000000ull, &sec, &usec);
	    f.accepted = parse_with_correction(ps, f.bits, sec, usec,
		    f.amplitude, &f);
	}
	if (accepted_only && !f.accepted)
	    continue;
	if (count < cap)
	    out[count] = f;
	count++;
    }
    return count;
}

size_t
oracle_main_loop(const float *m, size_t n, int flags, long long t0,
	unsigned long long rate, FILE *out) {
    /* planes-main-loop.c:49-99 with PLANES_INPUT_FROM_RAW */
    struct oracle_plane_set *ps = oracle_plane_set_new();
    int last_day = -1;
    size_t p, accepted = 0;
    char line[256];

    for (p = 0; n >= ORACLE_WINDOW && p + ORACLE_WINDOW <= n; p++) {
	char bits[ORACLE_MSG_BITS + 1];
	double a56, amplitude;
	long long sec, usec;

	if (!oracle_preamble(m, p))
	    continue;
	oracle_slice(m, p, bits, &a56);
	amplitude = a56 / 56;
	stream_time(p, t0, rate, &sec, &usec);
	{
	    struct oracle_frame info;
	    memset(&info, 0, sizeof info);
	    if (!parse_with_correction(ps, bits, sec, usec, amplitude, &info))
		continue;				/* :76-77 */
	}
	accepted++;

	if (flags & 0x20)				/* :79-82 */
	    fprintf(out, "%s @ %lld,%lld,%llu,%g\n", bits, sec, usec, 0ull,
		    amplitude);
	if (flags & 0x10) {				/* :84-94 */
	    struct tm tm;
	    time_t t = (time_t) sec;

	    localtime_r(&t, &tm);
	    if (last_day != tm.tm_yday) {
		fprintf(out, "--- %d-%02d-%02d ---\n", tm.tm_year + 1900,
			tm.tm_mon + 1, tm.tm_mday);
		last_day = tm.tm_yday;
	    }
	    oracle_print_last(ps, line, sizeof line);
	    fprintf(out, "%s\n", line);
	}
    }
    oracle_plane_set_delete(ps);
    return accepted;
}
