// "RHR1" read container (own format; stands in for SLOW5/BLOW5 until a native reader lands).
// The loader produces the SoA/CSR batch the C ABI takes: raw int16 samples + per-read calibration
// {offset (double), scale = (float)(range/digitisation)} exactly as ri_read_sig_slow5 derives them (rsig.c:494).
#include "rh_common.h"
#include <exception>

struct rh_reads_s {
	std::vector<std::string> names;
	std::vector<int16_t> samples;
	std::vector<uint64_t> offsets;
	std::vector<double> cal_offset;
	std::vector<float> cal_scale;
};

static long file_remaining(FILE *fp)
{
	const long at = ftell(fp);
	if (at < 0 || fseek(fp, 0, SEEK_END) != 0) return -1;
	const long end = ftell(fp);
	fseek(fp, at, SEEK_SET);
	return end < at ? -1 : end - at;
}

static rh_reads *reads_load_rhr(const char *path);
static rh_reads *reads_load_blow5(const char *path);

extern "C" rh_reads *rh_reads_load(const char *path)
{
	try {
		char magic[6] = {0};
		FILE *fp = fopen(path, "rb");
		if (!fp) { rh_set_error("cannot open %s", path); return 0; }
		const size_t got = fread(magic, 1, 6, fp);
		fclose(fp);
		if (got == 6 && !memcmp(magic, "BLOW5\1", 6)) return reads_load_blow5(path);
		return reads_load_rhr(path);
	}
	catch (const std::exception &e) { rh_set_error("%s: %s", path, e.what()); return 0; }   // (bad_alloc must not cross the C ABI)
}

static rh_reads *reads_load_rhr(const char *path)
{
	FILE *fp = fopen(path, "rb");
	if (!fp) { rh_set_error("cannot open %s", path); return 0; }
	char magic[4]; uint32_t n;
	if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, "RHR1", 4) != 0 || fread(&n, 4, 1, fp) != 1) {
		fclose(fp); rh_set_error("%s: not an RHR1 file", path); return 0;
	}
	rh_reads *r = new rh_reads_s();
	r->offsets.push_back(0);
	for (uint32_t i = 0; i < n; ++i) {
		uint32_t l, ns; double dig, range, off;
		std::string name;
		bool ok = fread(&l, 4, 1, fp) == 1 && (long)l <= file_remaining(fp);      // lengths are checked against what is left of the file
		if (ok) { name.resize(l); ok = l == 0 || fread(&name[0], 1, l, fp) == l; }
		ok = ok && fread(&ns, 4, 1, fp) == 1 && fread(&dig, 8, 1, fp) == 1 && fread(&range, 8, 1, fp) == 1 && fread(&off, 8, 1, fp) == 1;
		ok = ok && (long)ns * 2 <= file_remaining(fp);
		if (ok) {
			size_t o = r->samples.size();
			r->samples.resize(o + ns);
			ok = ns == 0 || fread(&r->samples[o], 2, ns, fp) == ns;
		}
		if (!ok) { fclose(fp); delete r; rh_set_error("%s: truncated at read %u", path, i); return 0; }
		r->names.push_back(name);
		r->offsets.push_back(r->samples.size());
		r->cal_offset.push_back(off);
		r->cal_scale.push_back((float)(range / dig));
	}
	fclose(fp);
	return r;
}

// ---------------------------------------------------------------------------------------------------- BLOW5
// Binary SLOW5 (hasindu2008/slow5lib, file format spec 1.0.0; what ri_read_sig_slow5 rsig.c:478-533 gets through slow5lib):
//   file header : "BLOW5\1" | version major, minor, patch (u8 x 3) | record compression (u8: 0 none, 1 zlib, 2 zstd)
//                 | signal compression (u8: 0 none, 1 svb-zd; files from version 0.2.0 on) | number of read groups (u32)
//                 | zero padding up to byte 64 | header text size (u32) | SLOW5 header text
//   record      : record size (u64) | body (deflated as a whole when record compression = zlib):
//                 read_id length (u16) | read_id | read_group (u32) | digitisation, offset, range, sampling_rate (f64 x 4)
//                 | len_raw_signal (u64) | raw_signal (i16 x len) | auxiliary fields (ignored here)
//   end of file : "5WOLB"
// Records are decoded straight into the SoA batch (int16 samples + per-read calibration): the raw->pA conversion is the
// device's.  zstd records and svb-zd signal compression are refused with an error (neither library is in this image).
#include <zlib.h>

static rh_reads *reads_load_blow5(const char *path)
{
	FILE *fp = fopen(path, "rb");
	if (!fp) { rh_set_error("cannot open %s", path); return 0; }
	auto fail = [&](rh_reads *r, const char *what) { fclose(fp); delete r; rh_set_error("%s: %s", path, what); return (rh_reads*)0; };
	unsigned char hd[64];
	if (fread(hd, 1, 64, fp) != 64 || memcmp(hd, "BLOW5\1", 6)) return fail(0, "not a BLOW5 file");
	const int vmaj = hd[6], vmin = hd[7];
	const int rec_comp = hd[9];
	const bool has_sig_byte = vmaj > 0 || vmin >= 2;
	const int sig_comp = has_sig_byte ? hd[10] : 0;
	if (rec_comp == 2) return fail(0, "zstd-compressed BLOW5 records are not supported (convert with `slow5tools view -c zlib`)");
	if (rec_comp != 0 && rec_comp != 1) return fail(0, "unknown BLOW5 record compression");
	if (sig_comp != 0) return fail(0, "svb-zd signal compression is not supported (convert with `slow5tools view -s none`)");
	uint32_t hsize;
	if (fread(&hsize, 4, 1, fp) != 1 || (long)hsize > file_remaining(fp) || fseek(fp, (long)hsize, SEEK_CUR)) return fail(0, "truncated BLOW5 header");
	rh_reads *r = new rh_reads_s();
	r->offsets.push_back(0);
	std::vector<unsigned char> raw, body;
	for (;;) {
		unsigned char szb[8];
		const size_t got = fread(szb, 1, 8, fp);
		if (got >= 5 && !memcmp(szb, "5WOLB", 5)) break;               // end-of-file marker
		if (got == 0) break;                                           // (files cut before the marker still give their records)
		if (got != 8) return fail(r, "truncated BLOW5 record");
		uint64_t rsz; memcpy(&rsz, szb, 8);
		if ((long)rsz > file_remaining(fp) || rsz > (1ull << 34)) return fail(r, "implausible BLOW5 record size");
		raw.resize(rsz);
		if (rsz && fread(raw.data(), 1, rsz, fp) != rsz) return fail(r, "truncated BLOW5 record");
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
				if (out == body.size()) body.resize(body.size() * 2);
				zs.next_out = body.data() + out; zs.avail_out = (uInt)(body.size() - out);
				rc = inflate(&zs, Z_NO_FLUSH);
				out = body.size() - zs.avail_out;
			} while (rc == Z_OK);
			inflateEnd(&zs);
			if (rc != Z_STREAM_END) return fail(r, "corrupt zlib BLOW5 record");
			b = body.data(); blen = out;
		}
		size_t at = 0;
		auto need = [&](size_t n) { return at + n <= blen; };
		uint16_t idl; uint32_t rg; double dig, off, range, rate; uint64_t ns;
		if (!need(2)) return fail(r, "short BLOW5 record"); memcpy(&idl, b + at, 2); at += 2;
		if (!need(idl)) return fail(r, "short BLOW5 record");
		std::string name((const char*)b + at, idl); at += idl;
		if (!need(4 + 32 + 8)) return fail(r, "short BLOW5 record");
		memcpy(&rg, b + at, 4); at += 4;
		memcpy(&dig, b + at, 8); at += 8; memcpy(&off, b + at, 8); at += 8; memcpy(&range, b + at, 8); at += 8; memcpy(&rate, b + at, 8); at += 8;
		memcpy(&ns, b + at, 8); at += 8;
		if (ns > (1ull << 32) || !need(ns * 2)) return fail(r, "short BLOW5 record (signal)");
		const size_t o = r->samples.size();
		r->samples.resize(o + ns);
		if (ns) memcpy(&r->samples[o], b + at, ns * 2);
		(void)rg; (void)rate;
		r->names.push_back(name);
		r->offsets.push_back(r->samples.size());
		r->cal_offset.push_back(off);
		r->cal_scale.push_back((float)(range / dig));                  // rsig.c:494
	}
	fclose(fp);
	return r;
}

// Writer (tests, format conversion): one read group, no auxiliary fields; zlib != 0 deflates every record.
extern "C" int rh_reads_write_blow5(const char *path, uint32_t n, const char *const *names, const int16_t *samples, const uint64_t *offsets,
                                    double digitisation, double range, double offset, double sampling_rate, int zlib_records)
{
	FILE *fp = fopen(path, "wb");
	if (!fp) { rh_set_error("cannot write %s", path); return -1; }
	unsigned char hd[64];
	memset(hd, 0, sizeof(hd));
	memcpy(hd, "BLOW5\1", 6);
	hd[6] = 1; hd[7] = 0; hd[8] = 0;                                   // file format 1.0.0
	hd[9] = zlib_records ? 1 : 0; hd[10] = 0;
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
		put(samples + offsets[i], ns * 2);
		const unsigned char *out = body.data();
		uint64_t rsz = body.size();
		if (zlib_records) {
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
extern "C" uint32_t rh_reads_n(const rh_reads *r) { return (uint32_t)r->names.size(); }
extern "C" const char *rh_reads_name(const rh_reads *r, uint32_t i) { return i < r->names.size() ? r->names[i].c_str() : 0; }

extern "C" int rh_reads_batch(const rh_reads *r, rh_read_batch_t *out)
{
	memset(out, 0, sizeof(*out));
	out->n_reads = (uint32_t)r->names.size();
	out->samples = r->samples.data();
	out->offsets = r->offsets.data();
	out->cal_offset = r->cal_offset.data();
	out->cal_scale = r->cal_scale.data();
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
