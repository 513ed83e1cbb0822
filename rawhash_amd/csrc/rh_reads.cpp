// Read containers -> the SoA/CSR batch the C ABI takes: raw int16 samples + per-read calibration {offset (double),
// scale = (float)(range/digitisation)} exactly as ri_read_sig_slow5 derives them (rsig.c:494).
//   BLOW5 (binary SLOW5): what the reference reads through slow5lib (rsig.c:170-206, 478-533) - uncompressed, zlib and zstd records,
//                         raw or svb-zd (StreamVByte zig-zag delta) signals
//   SLOW5 (text)        : the same records as tab-separated lines (slow5_open takes either, rsig.c:170-207): header lines (#..., @...), the column
//                         names on the last of them, raw_signal a comma-separated list
//   RHR1                : the repo's own trivial container (tests, the reference harness)
// Samples are decoded straight into ONE page-locked staging buffer (hipHostMalloc; plain memory when no HIP runtime answers, e.g. in
// the CPU test suite), so rh_map_batch / rh_map_submit upload them at PCIe speed without an intermediate copy.
#include "rh_common.h"
#include <algorithm>
#include <exception>
#include <dlfcn.h>
#include <sys/stat.h>
#include <thread>

extern "C" void *rh_pinned_alloc(size_t bytes);
extern "C" void rh_pinned_free(void *p);

namespace {
// growable int16 staging buffer in page-locked host memory
struct Staging {
	int16_t *p = nullptr; size_t n = 0, cap = 0; bool pinned = false;
	~Staging() { release(); }
	void release() { if (p) { if (pinned) rh_pinned_free(p); else free(p); } p = nullptr; n = cap = 0; }
	bool reserve(size_t want)
	{
		if (want <= cap) return true;
		size_t nc = cap ? cap : ((size_t)1 << 20);
		while (nc < want) nc *= 2;
		static const bool no_pin = getenv("RH_READS_NO_PIN") != nullptr;
		int16_t *q = no_pin ? nullptr : (int16_t*)rh_pinned_alloc(nc * 2);
		const bool pin = q != nullptr;
		if (!q) q = (int16_t*)malloc(nc * 2);
		if (!q) return false;
		if (n) memcpy(q, p, n * 2);
		if (p) { if (pinned) rh_pinned_free(p); else free(p); }
		p = q; cap = nc; pinned = pin;
		return true;
	}
	int16_t *grow(size_t k) { if (!reserve(n + k)) return nullptr; int16_t *at = p + n; n += k; return at; }
};
}

struct rh_reads_s {
	std::vector<std::string> names;
	Staging samples;
	std::vector<uint64_t> offsets;
	std::vector<double> cal_offset;
	std::vector<float> cal_scale;
	std::vector<uint32_t> n_filtered;      // samples of each read that pass the reader's pA filter (rsig.c:496-503): the read's l_sig, counted here on the host
};

// size of a regular file; anything else (a FIFO, a failed fstat) has no size to check lengths against: "unknown" = a bound no length
// field reaches, and the absolute caps below (name length, samples per read, decompressed record size) are what limits an allocation
static const uint64_t kSizeUnknown = 1ull << 62;
static const uint64_t kMaxNameLen = 1u << 16, kMaxReadSamples = 1ull << 32, kMaxRecordBytes = 1ull << 33;
static uint64_t file_size_of(FILE *fp)
{
	struct stat st;
	return fstat(fileno(fp), &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 0 ? (uint64_t)st.st_size : kSizeUnknown;
}

static rh_reads *reads_load_rhr(const char *path);
static rh_reads *reads_load_blow5(const char *path);
static rh_reads *reads_load_slow5_text(const char *path);

extern "C" rh_reads *rh_reads_load(const char *path)
{
	try {
		char magic[15] = {0};
		FILE *fp = fopen(path, "rb");
		if (!fp) { rh_set_error("cannot open %s", path); return 0; }
		const size_t got = fread(magic, 1, 14, fp);
		fclose(fp);
		rh_reads *r = (got >= 6 && !memcmp(magic, "BLOW5\1", 6)) ? reads_load_blow5(path) : (got == 14 && !memcmp(magic, "#slow5_version", 14)) ? reads_load_slow5_text(path) : reads_load_rhr(path);
		if (r) {	// what ri_read_sig knows when it returns (l_sig): lets rh_map_batch fetch only the signal its rounds consume
			rh_read_batch_t b;
			rh_reads_batch(r, &b);
			r->n_filtered.assign(b.n_reads ? b.n_reads : 1, 0);
			if (rh_count_filtered(&b, r->n_filtered.data(), 0)) { rh_reads_destroy(r); return 0; }
		}
		return r;
	}
	catch (const std::exception &e) { rh_set_error("%s: %s", path, e.what()); return 0; }   // (bad_alloc must not cross the C ABI)
}

static rh_reads *reads_load_rhr(const char *path)
{
	FILE *fp = fopen(path, "rb");
	if (!fp) { rh_set_error("cannot open %s", path); return 0; }
	const uint64_t fsize = file_size_of(fp);                        // once: positions are tracked arithmetically from here on
	uint64_t at = 0;
	char magic[4]; uint32_t n;
	if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, "RHR1", 4) != 0 || fread(&n, 4, 1, fp) != 1) {
		fclose(fp); rh_set_error("%s: not an RHR1 file", path); return 0;
	}
	at = 8;
	rh_reads *r = new rh_reads_s();
	r->offsets.push_back(0);
	for (uint32_t i = 0; i < n; ++i) {
		uint32_t l = 0, ns = 0; double dig = 0, range = 0, off = 0;
		std::string name;
		bool ok = fread(&l, 4, 1, fp) == 1 && (at += 4, at <= fsize && (uint64_t)l <= fsize - at && (uint64_t)l <= kMaxNameLen);      // lengths are checked against what is left of the file
		if (ok) { name.resize(l); ok = l == 0 || fread(&name[0], 1, l, fp) == l; at += l; }
		ok = ok && fread(&ns, 4, 1, fp) == 1 && fread(&dig, 8, 1, fp) == 1 && fread(&range, 8, 1, fp) == 1 && fread(&off, 8, 1, fp) == 1;
		at += 28;
		ok = ok && at <= fsize && (uint64_t)ns * 2 <= fsize - at;
		if (ok) {
			int16_t *dst = r->samples.grow(ns);
			ok = dst != nullptr && (ns == 0 || fread(dst, 2, ns, fp) == ns);
			at += (uint64_t)ns * 2;
		}
		if (!ok) { fclose(fp); delete r; rh_set_error("%s: truncated at read %u", path, i); return 0; }
		r->names.push_back(name);
		r->offsets.push_back(r->samples.n);
		r->cal_offset.push_back(off);
		r->cal_scale.push_back((float)(range / dig));
	}
	fclose(fp);
	return r;
}

// ---------------------------------------------------------------------------------------------------- SLOW5 (text)
// The ASCII form of the format slow5lib reads (what `slow5tools view` prints; ri_read_sig_slow5 takes the same fields from either form, rsig.c:478-533):
//   #slow5_version<TAB>x.y.z, #num_read_groups<TAB>n, @attribute lines, a line of column types (#char*<TAB>uint32_t ...), a line of column names
//   (#read_id<TAB>read_group<TAB>digitisation<TAB>offset<TAB>range<TAB>sampling_rate<TAB>len_raw_signal<TAB>raw_signal[<TAB>auxiliary fields]), then one
//   line per read with raw_signal as comma-separated integers.  Columns are found by name; auxiliary fields are skipped.
static rh_reads *reads_load_slow5_text(const char *path)
{
	FILE *fp = fopen(path, "rb");
	if (!fp) { rh_set_error("cannot open %s", path); return 0; }
	rh_reads *r = new rh_reads_s();
	r->offsets.push_back(0);
	auto fail = [&](const char *what, uint64_t line_no) -> rh_reads* { rh_set_error("%s: line %llu: %s", path, (unsigned long long)line_no, what); fclose(fp); delete r; return (rh_reads*)0; };
	char *line = nullptr; size_t cap = 0; ssize_t len;
	int c_id = -1, c_dig = -1, c_off = -1, c_rng = -1, c_len = -1, c_sig = -1, n_cols = 0;
	uint64_t line_no = 0;
	bool have_cols = false;
	struct Free { char *&p; ~Free() { free(p); } } free_line{line};
	while ((len = getline(&line, &cap, fp)) >= 0) {
		++line_no;
		while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
		if (len == 0) continue;
		if (line[0] == '@') continue;                                  // read-group attributes
		if (line[0] == '#') {
			if (!strncmp(line, "#read_id", 8) && (line[8] == '\t' || line[8] == 0)) {	// the column names
				int col = 0;
				for (char *tok = line + 1, *nx; tok; tok = nx, ++col) {
					nx = strchr(tok, '\t');
					if (nx) *nx++ = 0;
					if (!strcmp(tok, "read_id")) c_id = col; else if (!strcmp(tok, "digitisation")) c_dig = col; else if (!strcmp(tok, "offset")) c_off = col;
					else if (!strcmp(tok, "range")) c_rng = col; else if (!strcmp(tok, "len_raw_signal")) c_len = col; else if (!strcmp(tok, "raw_signal")) c_sig = col;
				}
				n_cols = col;
				if (c_id < 0 || c_dig < 0 || c_off < 0 || c_rng < 0 || c_len < 0 || c_sig < 0) return fail("a primary column (read_id, digitisation, offset, range, len_raw_signal, raw_signal) is missing", line_no);
				have_cols = true;
			}
			continue;                                                   // version, read groups, column types
		}
		if (!have_cols) return fail("a record before the column names", line_no);
		const char *f_id = nullptr, *f_sig = nullptr; double dig = 0, off = 0, rng = 0; uint64_t ns = 0;
		int col = 0;
		bool bad_num = false;
		auto num = [&](const char *t, double &v) { char *e; v = strtod(t, &e); if (e == t || *e) bad_num = true; };
		for (char *tok = line, *nx; tok; tok = nx, ++col) {
			nx = strchr(tok, '\t');
			if (nx) *nx++ = 0;
			if (col == c_id) f_id = tok; else if (col == c_sig) f_sig = tok;
			else if (col == c_dig) num(tok, dig); else if (col == c_off) num(tok, off); else if (col == c_rng) num(tok, rng);
			else if (col == c_len) { char *e; ns = strtoull(tok, &e, 10); if (e == tok || *e || *tok == '-') bad_num = true; }
		}
		// every primary column has to be there (a line cut short after raw_signal would otherwise map with offset / range 0: wrong calibration, no error)
		const int last_primary = std::max(std::max(std::max(c_id, c_dig), std::max(c_off, c_rng)), std::max(c_len, c_sig));
		if (col <= last_primary) return fail("fewer fields than the primary columns need", line_no);
		if (bad_num) return fail("digitisation / offset / range / len_raw_signal is not a number", line_no);
		if (!f_id || !f_sig) return fail("no read_id / raw_signal field", line_no);
		if (strlen(f_id) > kMaxNameLen || ns >= kMaxReadSamples || !(dig > 0)) return fail("implausible read (name length, signal length, digitisation)", line_no);
		int16_t *dst = r->samples.grow(ns);
		if (ns && !dst) return fail("out of memory", line_no);
		uint64_t k = 0;
		if (!(f_sig[0] == '.' && f_sig[1] == 0))
			for (const char *p = f_sig; *p;) {
				char *e;
				const long v = strtol(p, &e, 10);
				if (e == p) return fail("raw_signal is not a comma-separated list of integers", line_no);
				if (v < -32768 || v > 32767) return fail("a raw_signal value beyond int16", line_no);
				if (k < ns) dst[k] = (int16_t)v;
				++k;
				p = *e == ',' ? e + 1 : e;
				if (*e && *e != ',') return fail("raw_signal is not a comma-separated list of integers", line_no);
			}
		if (k != ns) return fail("raw_signal does not hold len_raw_signal values", line_no);
		r->names.push_back(f_id);
		r->offsets.push_back(r->samples.n);
		r->cal_offset.push_back(off);
		r->cal_scale.push_back((float)(rng / dig));
	}
	if (!have_cols) return fail("no column names (#read_id ...)", line_no);
	fclose(fp);
	return r;
}

// ---------------------------------------------------------------------------------------------------- BLOW5
// Binary SLOW5 (hasindu2008/slow5lib, file format spec 1.0.0; what ri_read_sig_slow5 rsig.c:478-533 gets through slow5lib).
// slow5lib is an empty submodule of the reference tree, so this follows the published format, and is pinned by byte-level fixtures
// the tests assemble by hand from it (tests/test_abi.py) - "parity unpinned" against slow5lib itself until a file written by
// slow5tools is available:
//   file header : "BLOW5\1" | version major, minor, patch (u8 x 3) | record compression (u8: 0 none, 1 zlib, 2 zstd)
//                 | signal compression (u8: 0 none, 1 svb-zd; files from version 0.2.0 on) | number of read groups (u32)
//                 | zero padding up to byte 64 | header text size (u32) | SLOW5 header text
//   record      : record size (u64) | body (compressed as a whole: one zlib stream / one zstd frame per record):
//                 read_id length (u16) | read_id | read_group (u32) | digitisation, offset, range, sampling_rate (f64 x 4)
//                 | len_raw_signal (u64) | raw_signal | auxiliary fields (ignored here)
//   raw_signal  : i16 x len, or with svb-zd (SLOW5 specification 1.0.0, "BLOW5 signal compression"; slow5lib slow5_rec_to_mem / slow5_press.c
//                 ptr_compress_svb_zd): compressed byte count (u64: slow5lib writes a size_t, 8 bytes on the 64-bit platforms it supports)
//                 | u32 number of values | StreamVByte block of the zig-zag-encoded first differences (x[0] - 0, x[1] - x[0], ...) of the
//                 samples widened to 32 bits; the count covers the u32 and the block.
//                 StreamVByte (Lemire's 32-bit format): ceil(n / 4) control bytes, two bits per value (bytes used - 1, first value in
//                 the low bits), then the values' 1 - 4 little-endian data bytes back to back.
//   end of file : "5WOLB"
// ONE layout is accepted for the svb-zd signal - the u64 count above - and anything else is refused with an error: the value count must
// equal len_raw_signal and the decode must consume exactly the counted bytes.  (Unpinned against slow5lib itself: no file written by
// slow5tools exists in this image)
#include <zlib.h>

namespace {
// libzstd is loaded at run time (the image has the library, not its headers): only the one-shot frame API is needed
struct Zstd {
	void *h = nullptr;
	size_t (*decompress)(void*, size_t, const void*, size_t) = nullptr;
	unsigned long long (*content_size)(const void*, size_t) = nullptr;
	unsigned (*is_error)(size_t) = nullptr;
	size_t (*compress)(void*, size_t, const void*, size_t, int) = nullptr;
	size_t (*bound)(size_t) = nullptr;
	bool ok = false;
	Zstd()
	{
		for (const char *nm : {"libzstd.so.1", "libzstd.so"}) if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
		if (!h) return;
		decompress = (decltype(decompress))dlsym(h, "ZSTD_decompress"); content_size = (decltype(content_size))dlsym(h, "ZSTD_getFrameContentSize");
		is_error = (decltype(is_error))dlsym(h, "ZSTD_isError"); compress = (decltype(compress))dlsym(h, "ZSTD_compress"); bound = (decltype(bound))dlsym(h, "ZSTD_compressBound");
		ok = decompress && content_size && is_error && compress && bound;
	}
};
Zstd &zstd() { static Zstd z; return z; }

// StreamVByte block of n values at p (at most avail bytes): decoded zig-zag deltas accumulated into int16 samples.
// Returns the bytes consumed, 0 if the block does not fit avail.
size_t svb_zd_decode(const unsigned char *p, size_t avail, uint32_t n, int16_t *out)
{
	const size_t n_ctl = ((size_t)n + 3) / 4;
	if (n_ctl > avail) return 0;
	const unsigned char *ctl = p, *dat = p + n_ctl, *end = p + avail;
	int32_t prev = 0;
	for (uint32_t i = 0; i < n; ++i) {
		const unsigned code = (ctl[i >> 2] >> ((i & 3) * 2)) & 3u;
		if (dat + code + 1 > end) return 0;
		uint32_t v = dat[0];
		if (code >= 1) v |= (uint32_t)dat[1] << 8;
		if (code >= 2) v |= (uint32_t)dat[2] << 16;
		if (code >= 3) v |= (uint32_t)dat[3] << 24;
		dat += code + 1;
		const int32_t d = (int32_t)(v >> 1) ^ -(int32_t)(v & 1u);     // zig-zag
		prev += d;
		out[i] = (int16_t)prev;
	}
	return (size_t)(dat - p);
}

size_t svb_zd_encode(const int16_t *x, uint32_t n, std::vector<unsigned char> &out)   // (the writer: tests, format conversion)
{
	const size_t n_ctl = ((size_t)n + 3) / 4, at0 = out.size();
	out.resize(at0 + n_ctl, 0);
	int32_t prev = 0;
	for (uint32_t i = 0; i < n; ++i) {
		const int32_t d = (int32_t)x[i] - prev; prev = x[i];
		const uint32_t v = ((uint32_t)d << 1) ^ (uint32_t)(d >> 31);
		const unsigned code = v < (1u << 8) ? 0 : v < (1u << 16) ? 1 : v < (1u << 24) ? 2 : 3;
		out[at0 + (i >> 2)] |= (unsigned char)(code << ((i & 3) * 2));
		for (unsigned b = 0; b <= code; ++b) out.push_back((unsigned char)(v >> (8 * b)));
	}
	return out.size() - at0;
}
}

static rh_reads *reads_load_blow5(const char *path)
{
	FILE *fp = fopen(path, "rb");
	if (!fp) { rh_set_error("cannot open %s", path); return 0; }
	auto fail = [&](rh_reads *r, const char *what) { fclose(fp); delete r; rh_set_error("%s: %s", path, what); return (rh_reads*)0; };
	const uint64_t fsize = file_size_of(fp);
	unsigned char hd[64];
	if (fread(hd, 1, 64, fp) != 64 || memcmp(hd, "BLOW5\1", 6)) return fail(0, "not a BLOW5 file");
	const int vmaj = hd[6], vmin = hd[7];
	const int rec_comp = hd[9];
	const bool has_sig_byte = vmaj > 0 || vmin >= 2;
	const int sig_comp = has_sig_byte ? hd[10] : 0;
	if (rec_comp < 0 || rec_comp > 2) return fail(0, "unknown BLOW5 record compression");
	if (rec_comp == 2 && !zstd().ok) return fail(0, "zstd-compressed BLOW5 records need libzstd.so.1, which could not be loaded (convert with `slow5tools view -c zlib`)");
	if (sig_comp != 0 && sig_comp != 1) return fail(0, "unknown BLOW5 signal compression");
	uint32_t hsize;
	uint64_t at = 64;
	if (fread(&hsize, 4, 1, fp) != 1 || (at += 4, (uint64_t)hsize > fsize - at) || fseek(fp, (long)hsize, SEEK_CUR)) return fail(0, "truncated BLOW5 header");
	at += hsize;
	rh_reads *r = new rh_reads_s();
	r->offsets.push_back(0);
	std::vector<unsigned char> raw, body;
	for (;;) {
		unsigned char szb[8];
		const size_t got = fread(szb, 1, 8, fp);
		if (got >= 5 && !memcmp(szb, "5WOLB", 5)) break;               // end-of-file marker
		if (got == 0) break;                                           // (files cut before the marker still give their records)
		if (got != 8) return fail(r, "truncated BLOW5 record");
		at += 8;
		uint64_t rsz; memcpy(&rsz, szb, 8);
		if (at > fsize || rsz > fsize - at || rsz >= (1ull << 32)) return fail(r, "implausible BLOW5 record size");
		// a decompressed record: deflate expands at most ~1032-fold (a long flat signal does compress that well), zstd is bounded by the record limit; records are
		// one read - far below 4 GiB, which is also what zlib's 32-bit counters take
		const uint64_t body_cap = rsz * 1040u + (1u << 20) < kMaxRecordBytes ? rsz * 1040u + (1u << 20) : kMaxRecordBytes;
		raw.resize(rsz);
		if (rsz && fread(raw.data(), 1, rsz, fp) != rsz) return fail(r, "truncated BLOW5 record");
		at += rsz;
		const unsigned char *b = raw.data();
		size_t blen = rsz;
		if (rec_comp == 1) {	// one zlib stream per record
			z_stream zs; memset(&zs, 0, sizeof(zs));
			if (inflateInit(&zs) != Z_OK) return fail(r, "zlib init failed");
			body.resize(rsz * 4 + 1024);
			zs.next_in = raw.data(); zs.avail_in = (uInt)rsz;
			size_t out = 0;
			int rc;
			do {
				if (out == body.size()) { if (body.size() >= body_cap) { inflateEnd(&zs); return fail(r, "corrupt zlib BLOW5 record (implausible expansion)"); } body.resize(body.size() * 2 < body_cap ? body.size() * 2 : body_cap); }
				const size_t room = body.size() - out;
				zs.next_out = body.data() + out; zs.avail_out = (uInt)(room < 0x40000000u ? room : 0x40000000u);
				const uInt gave = zs.avail_out;
				rc = inflate(&zs, Z_NO_FLUSH);
				out += gave - zs.avail_out;
			} while (rc == Z_OK);
			inflateEnd(&zs);
			if (rc != Z_STREAM_END) return fail(r, "corrupt zlib BLOW5 record");
			b = body.data(); blen = out;
		} else if (rec_comp == 2) {	// one zstd frame per record
			const unsigned long long full = zstd().content_size(raw.data(), rsz);
			if (full == 0ull - 1 || full == 0ull - 2 || full > body_cap) return fail(r, "corrupt zstd BLOW5 record (frame size)");   // (checked before anything is allocated for it)
			body.resize((size_t)full + 8);
			const size_t got2 = zstd().decompress(body.data(), body.size(), raw.data(), rsz);
			if (zstd().is_error(got2) || got2 != full) return fail(r, "corrupt zstd BLOW5 record");
			b = body.data(); blen = got2;
		}
		size_t pos = 0;
		auto need = [&](size_t n) { return pos + n <= blen; };
		uint16_t idl; uint32_t rg; double dig, off, range, rate; uint64_t ns;
		if (!need(2)) return fail(r, "short BLOW5 record"); memcpy(&idl, b + pos, 2); pos += 2;
		if (!need(idl)) return fail(r, "short BLOW5 record");
		std::string name((const char*)b + pos, idl); pos += idl;
		if (!need(4 + 32 + 8)) return fail(r, "short BLOW5 record");
		memcpy(&rg, b + pos, 4); pos += 4;
		memcpy(&dig, b + pos, 8); pos += 8; memcpy(&off, b + pos, 8); pos += 8; memcpy(&range, b + pos, 8); pos += 8; memcpy(&rate, b + pos, 8); pos += 8;
		memcpy(&ns, b + pos, 8); pos += 8;
		// the sample count is checked against what the record holds BEFORE the (page-locked) staging buffer grows for it
		if (ns > kMaxReadSamples) return fail(r, "short BLOW5 record (signal)");
		if (sig_comp == 0) {
			if (ns > (blen - pos) / 2) return fail(r, "short BLOW5 record (signal)");
			int16_t *dst = r->samples.grow(ns);
			if (!dst) return fail(r, "out of memory for the sample staging buffer");
			if (ns) memcpy(dst, b + pos, ns * 2);
		} else {	// svb-zd: u64 compressed byte count | u32 values | StreamVByte block (see the format notes above)
			uint64_t cnt = 0;
			if (!need(8 + 4)) return fail(r, "short BLOW5 record (svb-zd signal)");
			memcpy(&cnt, b + pos, 8);
			if (cnt < 4 || cnt > blen - pos - 8) return fail(r, "corrupt svb-zd signal in a BLOW5 record: the compressed byte count (u64 in front of the signal) does not fit the record");
			uint32_t nv; memcpy(&nv, b + pos + 8, 4);
			if (nv != ns) return fail(r, "corrupt svb-zd signal in a BLOW5 record: value count differs from len_raw_signal");
			if (ns > cnt - 4) return fail(r, "corrupt svb-zd signal in a BLOW5 record: more values than data bytes");   // (a value takes at least one data byte)
			int16_t *dst = r->samples.grow(ns);
			if (!dst) return fail(r, "out of memory for the sample staging buffer");
			const size_t avail = (size_t)cnt - 4, used = svb_zd_decode(b + pos + 12, avail, nv, dst);
			if ((ns && !used) || used != avail) return fail(r, "corrupt svb-zd signal in a BLOW5 record: the StreamVByte block does not end with the counted bytes");
		}
		(void)rg; (void)rate;
		r->names.push_back(name);
		r->offsets.push_back(r->samples.n);
		r->cal_offset.push_back(off);
		r->cal_scale.push_back((float)(range / dig));                  // rsig.c:494
	}
	fclose(fp);
	return r;
}

// Writer (tests, format conversion): one read group, no auxiliary fields.  `compression` & 0xFF = record compression (0 none, 1 zlib,
// 2 zstd); bit 8 = svb-zd signal compression (compressed byte count written as u64).
extern "C" int rh_reads_write_blow5(const char *path, uint32_t n, const char *const *names, const int16_t *samples, const uint64_t *offsets,
                                    double digitisation, double range, double offset, double sampling_rate, int compression)
{
	const int rec_comp = compression & 0xFF; const bool svb = (compression & 0x100) != 0;
	if (rec_comp > 2 || (rec_comp == 2 && !zstd().ok)) { rh_set_error("BLOW5 writer: record compression %d is not available", rec_comp); return -1; }
	FILE *fp = fopen(path, "wb");
	if (!fp) { rh_set_error("cannot write %s", path); return -1; }
	unsigned char hd[64];
	memset(hd, 0, sizeof(hd));
	memcpy(hd, "BLOW5\1", 6);
	hd[6] = 1; hd[7] = 0; hd[8] = 0;                                   // file format 1.0.0
	hd[9] = (unsigned char)rec_comp; hd[10] = svb ? 1 : 0;
	const uint32_t n_rg = 1;
	memcpy(hd + 11, &n_rg, 4);
	fwrite(hd, 1, 64, fp);
	const std::string text = "#slow5_version\t1.0.0\n#num_read_groups\t1\n"
	                         "#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n"
	                         "#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n";
	const uint32_t hsize = (uint32_t)text.size();
	fwrite(&hsize, 4, 1, fp);
	fwrite(text.data(), 1, text.size(), fp);
	std::vector<unsigned char> body, comp;
	for (uint32_t i = 0; i < n; ++i) {
		const uint16_t idl = (uint16_t)strlen(names[i]);
		const uint64_t ns = offsets[i + 1] - offsets[i];
		const uint32_t rg = 0;
		body.clear();
		auto put = [&](const void *p, size_t k) { const unsigned char *q = (const unsigned char*)p; body.insert(body.end(), q, q + k); };
		put(&idl, 2); put(names[i], idl); put(&rg, 4); put(&digitisation, 8); put(&offset, 8); put(&range, 8); put(&sampling_rate, 8); put(&ns, 8);
		if (!svb) put(samples + offsets[i], ns * 2);
		else {
			std::vector<unsigned char> blk;
			const uint32_t nv = (uint32_t)ns;
			blk.insert(blk.end(), (const unsigned char*)&nv, (const unsigned char*)&nv + 4);
			svb_zd_encode(samples + offsets[i], nv, blk);
			const uint64_t cnt = blk.size();
			put(&cnt, 8); put(blk.data(), blk.size());
		}
		const unsigned char *out = body.data();
		uint64_t rsz = body.size();
		if (rec_comp == 2) {
			comp.resize(zstd().bound(body.size()));
			const size_t cl = zstd().compress(comp.data(), comp.size(), body.data(), body.size(), 3);
			if (zstd().is_error(cl)) { fclose(fp); rh_set_error("zstd failed"); return -1; }
			out = comp.data(); rsz = cl;
		} else if (rec_comp == 1) {
			uLongf cl = compressBound((uLong)body.size());
			comp.resize(cl);
			if (compress2(comp.data(), &cl, body.data(), (uLong)body.size(), Z_DEFAULT_COMPRESSION) != Z_OK) { fclose(fp); rh_set_error("zlib failed"); return -1; }
			out = comp.data(); rsz = cl;
		}
		fwrite(&rsz, 8, 1, fp);
		fwrite(out, 1, rsz, fp);
	}
	fwrite("5WOLB", 1, 5, fp);
	{ const bool bad = ferror(fp) != 0; if (fclose(fp) != 0 || bad) { rh_set_error("%s: write failed (disk full?)", path); return -1; } }
	return 0;
}

extern "C" void rh_reads_destroy(rh_reads *r) { delete r; }
extern "C" int rh_reads_pinned(const rh_reads *r) { return r->samples.pinned ? 1 : 0; }
extern "C" uint32_t rh_reads_n(const rh_reads *r) { return (uint32_t)r->names.size(); }
extern "C" const char *rh_reads_name(const rh_reads *r, uint32_t i) { return i < r->names.size() ? r->names[i].c_str() : 0; }

extern "C" int rh_reads_batch(const rh_reads *r, rh_read_batch_t *out)
{
	memset(out, 0, sizeof(*out));
	out->n_reads = (uint32_t)r->names.size();
	out->samples = r->samples.p;
	out->offsets = r->offsets.data();
	out->cal_offset = r->cal_offset.data();
	out->cal_scale = r->cal_scale.data();
	out->n_filtered = r->n_filtered.empty() ? nullptr : r->n_filtered.data();
	return 0;
}

// The reader's filter as a count (rsig.c:496-503 / :363-374: pA = (raw + offset) * scale kept iff 30 < pA < 200; the same arithmetic as
// raw_to_pa of the device, rh_kernels.hip): out[r] = l_sig of read r.  n_threads <= 0: as many as the host has, at most 32.
extern "C" int rh_count_filtered(const rh_read_batch_t *in, uint32_t *out, int n_threads)
{
	if (in->samples_on_device) { rh_set_error("rh_count_filtered: host batches only"); return -1; }
	const uint32_t R = in->n_reads;
	if (n_threads <= 0) { n_threads = (int)std::thread::hardware_concurrency(); if (n_threads > 32) n_threads = 32; if (n_threads < 1) n_threads = 1; }
	if ((uint32_t)n_threads > R) n_threads = R ? (int)R : 1;
	auto work = [&](uint32_t r0, uint32_t r1) {
		for (uint32_t r = r0; r < r1; ++r) {
			const int16_t *p = in->samples + in->offsets[r];
			const uint64_t n = in->offsets[r + 1] - in->offsets[r];
			const double co = in->cal_offset ? in->cal_offset[r] : 0.0;
			const float cs = in->cal_scale ? in->cal_scale[r] : 1.0f;
			uint32_t k = 0;
			if (in->fast5_ingest) { const float fo = (float)co; for (uint64_t i = 0; i < n; ++i) { const float pa = ((float)p[i] + fo) * cs; k += (pa > 30.0f && pa < 200.0f) ? 1u : 0u; } }
			else for (uint64_t i = 0; i < n; ++i) { const float pa = (float)(((double)p[i] + co) * (double)cs); k += (pa > 30.0f && pa < 200.0f) ? 1u : 0u; }
			out[r] = k;
		}
	};
	std::vector<std::thread> th;
	// (contiguous blocks of reads of about equal sample counts)
	const uint64_t tot = R ? in->offsets[R] - in->offsets[0] : 0;
	uint32_t r0 = 0;
	for (int t = 0; t < n_threads; ++t) {
		uint32_t r1 = r0;
		const uint64_t upto = in->offsets ? in->offsets[0] + tot * (uint64_t)(t + 1) / (uint64_t)n_threads : 0;
		while (r1 < R && (t == n_threads - 1 || in->offsets[r1 + 1] <= upto)) ++r1;
		if (r1 > r0) th.emplace_back(work, r0, r1);
		r0 = r1;
	}
	for (auto &x : th) x.join();
	return 0;
}

extern "C" int rh_reads_write(const char *path, uint32_t n, const char *const *names, const int16_t *samples,
                              const uint64_t *offsets, double digitisation, double range, double offset)
{
	FILE *fp = fopen(path, "wb");
	if (!fp) { rh_set_error("cannot write %s", path); return -1; }
	fwrite("RHR1", 1, 4, fp);
	fwrite(&n, 4, 1, fp);
	for (uint32_t i = 0; i < n; ++i) {
		uint32_t l = (uint32_t)strlen(names[i]), ns = (uint32_t)(offsets[i + 1] - offsets[i]);
		fwrite(&l, 4, 1, fp); fwrite(names[i], 1, l, fp);
		fwrite(&ns, 4, 1, fp);
		fwrite(&digitisation, 8, 1, fp); fwrite(&range, 8, 1, fp); fwrite(&offset, 8, 1, fp);
		fwrite(samples + offsets[i], 2, ns, fp);
	}
	{ const bool bad = ferror(fp) != 0; if (fclose(fp) != 0 || bad) { rh_set_error("%s: write failed (disk full?)", path); return -1; } }
	return 0;
}
