// Index I/O and construction on the host (feeds the device-resident table; not a kernel target itself).
//   rh_index_load   : `.ind` reader, format of ri_idx_dump / ri_idx_load (reference rindex.c:545-648 / :650-776)
//   rh_index_build  : FASTA + k-mer model -> sketches -> keys/positions (ri_idx_gen rindex.c:900, worker_pipeline :100,
//                     ri_seq_to_sig rsig.c:13, load_pore rutils.c:133, worker_post rindex.c:311) and `.ind` writer
//   rh_mapopt_update: mid_occ calibration (ri_idx_cal_max_occ rindex.c:1018, ri_mapopt_update :1041)
// The `.ind` files written here are byte-identical to the reference's (key/value pairs in khash slot order, KhashOrder below)
// except for the 16 bytes of the header where the reference dumps two heap pointers of its ri_pore_t (written as 0 here).
#include "rh_index.h"
#include "rh_core.h"
#include <algorithm>
#include <exception>
#include <cmath>
#include <memory>
#include <thread>
#include <zlib.h>

namespace {

const int kBucketBits = 14;   // rindex.c:669: not serialised, always 14

struct Entry { uint32_t hash; uint32_t n; uint64_t val; };

bool rd(FILE *fp, void *dst, size_t sz, size_t n) { return fread(dst, sz, n, fp) == n; }

struct HostSeed { uint32_t hash; uint64_t y; };

struct SeedSink {
	std::vector<HostSeed> *out;
	void operator()(uint64_t x, uint64_t y) { out->push_back(HostSeed{(uint32_t)(x >> 6), y}); }
};

// nucleotide -> 2 bits, 4 = ambiguous (what seq_nt4_table encodes)
inline int nt4(unsigned char c)
{
	switch (c) {
		case 'A': case 'a': return 0;
		case 'C': case 'c': return 1;
		case 'G': case 'g': return 2;
		case 'T': case 't': case 'U': case 'u': return 3;
		default: return 4;
	}
}

// expected signal of a sequence: one (normalised) model level per k-mer, ambiguous bases repeat the previous k-mer
// (ri_seq_to_sig rsig.c:13-41); strand 1 walks the reverse complement
void seq_to_levels(const std::string &seq, const std::vector<float> &model, int k, int strand, std::vector<float> &out)
{
	const int len = (int)seq.size();
	const uint64_t mask = (1ULL << (2 * k)) - 1;
	uint64_t kmer = 0;
	out.clear();
	for (int i = 0; i < len; ++i) {
		const int pos = strand ? len - i - 1 : i;
		const int c = nt4((unsigned char)seq[pos]);
		if (c < 4) kmer = ((kmer << 2) | (uint64_t)(strand ? (3 ^ c) : c)) & mask;
		if (i + 1 < k) continue;
		out.push_back(model[kmer]);
	}
}

bool load_model(const char *path, int k, int lev_col, std::vector<float> &vals)
{
	FILE *fp = fopen(path, "r");
	if (!fp) { rh_set_error("cannot open pore model %s", path); return false; }
	const size_t n_expected = (size_t)1 << (2 * k);
	vals.assign(n_expected, 0.0f);
	char line[1024];
	size_t i = 0;
	double sum = 0, sum2 = 0;
	while (fgets(line, sizeof(line), fp)) {
		if (!strncmp(line, "kmer", 4)) continue;
		char *tok = line;
		for (int col = 0; tok && col < lev_col; ++col) { tok = strchr(tok, '\t'); if (tok) ++tok; }
		float v;
		if (!tok || sscanf(tok, "%f", &v) != 1) { fclose(fp); rh_set_error("pore model %s: cannot parse line %zu", path, i + 1); return false; }
		if (i < n_expected) vals[i] = v;
		sum += v; sum2 += v * v;
		++i;
	}
	fclose(fp);
	if (i == 0) { rh_set_error("pore model %s is empty", path); return false; }
	const double mean = sum / i, sd = sqrt(sum2 / i - mean * mean);
	for (size_t j = 0; j < i && j < n_expected; ++j) vals[j] = (vals[j] - mean) / sd;
	return true;
}

uint32_t revcomp_kmer(uint32_t x, int k)
{
	uint32_t y = 0;
	for (int i = 0; i < k; ++i) { y = (y << 2) | ((x & 3) ^ 3); x >>= 2; }
	return y;
}

// the (value, k-mer, revcomp k-mer) table stored next to the model in the file header (create_sorted_pairs rutils.c:88-115)
void make_pore_inds(const std::vector<float> &vals, int k, std::vector<unsigned char> &blob)
{
	struct P { float v; uint32_t ind, rev; };
	std::vector<P> p(vals.size());
	double sum = 0, sum2 = 0;
	for (float v : vals) { sum += v; sum2 += v * v; }
	const double mean = sum / vals.size(), sd = sqrt(sum2 / vals.size() - mean * mean);
	for (uint32_t i = 0; i < vals.size(); ++i) p[i] = P{(float)((vals[i] - mean) / sd), i, revcomp_kmer(i, k)};
	std::stable_sort(p.begin(), p.end(), [](const P &a, const P &b) { return a.v < b.v; });
	blob.resize(p.size() * sizeof(P));
	memcpy(blob.data(), p.data(), blob.size());
}

bool read_fasta(const char *path, std::vector<std::string> &names, std::vector<std::string> &seqs)
{
	gzFile fp = gzopen(path, "r");
	if (!fp) { rh_set_error("cannot open %s", path); return false; }
	std::vector<char> buf(1 << 16);
	std::string cur;
	bool in_header = false, have = false;
	std::string header;
	int n;
	while ((n = gzread(fp, buf.data(), (unsigned)buf.size())) > 0) {
		for (int i = 0; i < n; ++i) {
			const char c = buf[i];
			if (in_header) {
				if (c == '\n') {
					in_header = false;
					size_t e = header.find_first_of(" \t\r");
					names.push_back(header.substr(0, e));
					seqs.emplace_back();
					have = true;
				} else header.push_back(c);
			} else if (c == '>') { in_header = true; header.clear(); }
			else if (have && c != '\n' && c != '\r' && c != ' ' && c != '\t') seqs.back().push_back(c);
		}
	}
	gzclose(fp);
	if (names.empty()) { rh_set_error("%s: no FASTA records", path); return false; }
	return true;
}

void finalize_keys(rh_index_s &ix, std::vector<HostSeed> &all, int n_threads)
{
	// sort by (hash, y): per key the position list comes out ascending, as radix_sort_64 leaves it (rindex.c:350)
	auto less = [](const HostSeed &a, const HostSeed &b) { return a.hash != b.hash ? a.hash < b.hash : a.y < b.y; };
	if (n_threads > 1 && all.size() > (1u << 20)) {
		// split by the top byte of the hash, sort the 256 ranges concurrently
		std::vector<size_t> cnt(257, 0);
		for (const HostSeed &s : all) ++cnt[(s.hash >> 24) + 1];
		for (int i = 0; i < 256; ++i) cnt[i + 1] += cnt[i];
		std::vector<HostSeed> tmp(all.size());
		{ std::vector<size_t> w(cnt.begin(), cnt.end() - 1); for (const HostSeed &s : all) tmp[w[s.hash >> 24]++] = s; }
		all.swap(tmp);
		std::vector<std::thread> th;
		for (int t = 0; t < n_threads; ++t)
			th.emplace_back([&, t]() { for (int b = t; b < 256; b += n_threads) std::sort(all.begin() + cnt[b], all.begin() + cnt[b + 1], less); });
		for (auto &t : th) t.join();
	} else std::sort(all.begin(), all.end(), less);
	ix.key_hash.clear(); ix.key_n.clear(); ix.key_val.clear(); ix.pos.clear();
	for (size_t i = 0; i < all.size();) {
		size_t j = i + 1;
		while (j < all.size() && all[j].hash == all[i].hash) ++j;
		ix.key_hash.push_back(all[i].hash);
		ix.key_n.push_back((uint32_t)(j - i));
		if (j - i == 1) ix.key_val.push_back(all[i].y);
		else {
			ix.key_val.push_back(ix.pos.size());
			for (size_t t = i; t < j; ++t) ix.pos.push_back(all[t].y);
		}
		i = j;
	}
}

// Slot order of klib's khash (khash.h:232-330) for a bucket's keys inserted the way worker_post does (rindex.c:311-363:
// kh_resize(n_keys), then kh_put in ascending hash order): ri_idx_dump (rindex.c:545) writes the pairs in slot order, so a
// byte-identical .ind needs the table's geometry - power-of-two size, 0.77 load bound, hash = key >> 1, triangular probing,
// and the in-place "kick-out" rehash when the table doubles.  Insert-only: no deleted slots ever exist outside a rehash.
struct KhashOrder {
	uint32_t n_buckets = 0, size = 0, upper = 0;
	std::vector<uint8_t> used;
	std::vector<uint64_t> keys, vals;
	static uint32_t roundup32(uint32_t x) { --x; x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16; return ++x; }
	void resize(uint32_t want)
	{
		uint32_t nn = roundup32(want);
		if (nn < 4) nn = 4;
		if (size >= (uint32_t)(nn * 0.77 + 0.5)) return;             // requested size is too small
		std::vector<uint8_t> nused(nn, 0);
		if (n_buckets < nn) { keys.resize(nn); vals.resize(nn); }
		const uint32_t nmask = nn - 1;
		for (uint32_t j = 0; j != n_buckets; ++j) {
			if (!used[j]) continue;
			uint64_t key = keys[j], val = vals[j];
			used[j] = 0;
			for (;;) {                                                   // kick-out process
				uint32_t i = (uint32_t)(key >> 1) & nmask, step = 0;
				while (nused[i]) i = (i + (++step)) & nmask;
				nused[i] = 1;
				if (i < n_buckets && used[i]) { std::swap(keys[i], key); std::swap(vals[i], val); used[i] = 0; }
				else { keys[i] = key; vals[i] = val; break; }
			}
		}
		if (n_buckets > nn) { keys.resize(nn); vals.resize(nn); }
		used.swap(nused);
		n_buckets = nn;
		upper = (uint32_t)(n_buckets * 0.77 + 0.5);
	}
	uint32_t put(uint64_t key)
	{
		if (size >= upper) resize(n_buckets > (size << 1) ? n_buckets - 1 : n_buckets + 1);
		const uint32_t mask = n_buckets - 1;
		uint32_t i = (uint32_t)(key >> 1) & mask, step = 0;
		while (used[i]) i = (i + (++step)) & mask;
		used[i] = 1; keys[i] = key; ++size;
		return i;
	}
};

bool write_ind(const rh_index_s &ix, const char *path)
{
	FILE *fp = fopen(path, "wb");
	if (!fp) { rh_set_error("cannot write %s", path); return false; }
	const uint32_t pars[7] = {(uint32_t)ix.w, (uint32_t)ix.e, (uint32_t)ix.n, (uint32_t)ix.q, (uint32_t)ix.k, (uint32_t)ix.names.size(), (uint32_t)ix.flag};
	fwrite("RI", 1, 2, fp);
	fwrite(pars, 4, 7, fp);
	fwrite(&ix.diff, 4, 1, fp); fwrite(&ix.fine_min, 4, 1, fp); fwrite(&ix.fine_max, 4, 1, fp); fwrite(&ix.fine_range, 4, 1, fp);
	// raw ri_pore_t image (rutils.h:26): two pointers (meaningless on disk, written as 0), n_pore_vals, k, max_val, min_val
	unsigned char pore[32];
	memset(pore, 0, sizeof(pore));
	const float max_val = -5000.0f, min_val = 5000.0f;   // main.cpp:546-547, never updated afterwards
	memcpy(pore + 16, &ix.n_pore_vals, 4); memcpy(pore + 20, &ix.pore_k, 2); memcpy(pore + 24, &max_val, 4); memcpy(pore + 28, &min_val, 4);
	fwrite(pore, 1, 32, fp);
	fwrite(ix.pore_vals.data(), 4, ix.n_pore_vals, fp);
	fwrite(ix.pore_inds.data(), 1, (size_t)ix.n_pore_vals * 12, fp);
	for (size_t i = 0; i < ix.names.size(); ++i) {
		const uint8_t l = (uint8_t)ix.names[i].size();   // strlen narrowed to uint8 in the reference as well
		fwrite(&l, 1, 1, fp);
		fwrite(ix.names[i].data(), 1, l, fp);
		fwrite(&ix.lens[i], 4, 1, fp);
		if (ix.flag & RH_I_STORE_SIG) {	// rindex.c:590-598
			const std::vector<float> empty;
			const std::vector<float> &F = i < ix.sigF.size() ? ix.sigF[i] : empty;
			const uint32_t fl = (uint32_t)F.size();
			fwrite(&fl, 4, 1, fp); fwrite(F.data(), 4, fl, fp);
			if (!(ix.flag & RH_I_NO_REV_TARGET)) {
				const std::vector<float> &R = i < ix.sigR.size() ? ix.sigR[i] : empty;
				const uint32_t rl = (uint32_t)R.size();
				fwrite(&rl, 4, 1, fp); fwrite(R.data(), 4, rl, fp);
			}
		}
	}
	// regroup the hash-sorted keys by their low 14 bits (stable: hash order is kept inside a bucket)
	const uint32_t nb = 1u << kBucketBits, bmask = nb - 1;
	std::vector<uint64_t> start(nb + 1, 0);
	for (uint32_t h : ix.key_hash) ++start[(h & bmask) + 1];
	for (uint32_t b = 0; b < nb; ++b) start[b + 1] += start[b];
	std::vector<uint32_t> order(ix.key_hash.size());
	{ std::vector<uint64_t> w(start.begin(), start.end() - 1); for (uint32_t i = 0; i < ix.key_hash.size(); ++i) order[w[ix.key_hash[i] & bmask]++] = i; }
	std::vector<uint64_t> p, kv;
	for (uint32_t b = 0; b < nb; ++b) {
		p.clear(); kv.clear();
		KhashOrder kh;
		const uint32_t n_keys = (uint32_t)(start[b + 1] - start[b]);
		if (n_keys) kh.resize(n_keys);                               // worker_post: kh_resize(idx, h, n_keys) before the puts
		for (uint64_t t = start[b]; t < start[b + 1]; ++t) {
			const uint32_t i = order[t], n = ix.key_n[i];
			const uint64_t key = (uint64_t)(ix.key_hash[i] >> kBucketBits) << 1;
			const uint32_t slot = kh.put(key);
			if (n == 1) { kh.keys[slot] = key | 1; kh.vals[slot] = ix.key_val[i]; }
			else {
				kh.vals[slot] = (uint64_t)p.size() << 32 | n;
				p.insert(p.end(), ix.pos.begin() + ix.key_val[i], ix.pos.begin() + ix.key_val[i] + n);
			}
		}
		for (uint32_t sl = 0; sl < kh.n_buckets; ++sl) if (kh.used[sl]) { kv.push_back(kh.keys[sl]); kv.push_back(kh.vals[sl]); }   // ri_idx_dump: slot order
		const int32_t np = (int32_t)p.size();
		const uint32_t size = (uint32_t)(kv.size() / 2);
		fwrite(&np, 4, 1, fp);
		fwrite(p.data(), 8, p.size(), fp);
		fwrite(&size, 4, 1, fp);
		fwrite(kv.data(), 8, kv.size(), fp);
	}
	// a full disk must not leave a silently truncated index behind: stdio keeps the first write error in the stream
	const bool bad = ferror(fp) != 0;
	if (fclose(fp) != 0 || bad) { rh_set_error("%s: write failed (disk full?)", path); return false; }
	return true;
}

} // namespace

bool rh_load_model(const char *path, int k, int lev_col, std::vector<float> &vals) { return load_model(path, k, lev_col, vals); }
void rh_make_pore_inds(const std::vector<float> &vals, int k, std::vector<unsigned char> &blob) { make_pore_inds(vals, k, blob); }
bool rh_read_fasta(const char *path, std::vector<std::string> &names, std::vector<std::string> &seqs) { return read_fasta(path, names, seqs); }

static rh_index *index_load(const char *path);
extern "C" rh_index *rh_index_load(const char *path)
{
	try { return index_load(path); }
	catch (const std::exception &e) { rh_set_error("%s: %s", path, e.what()); return nullptr; }   // (bad_alloc must not cross the C ABI)
}

static rh_index *index_load(const char *path)
{
	FILE *fp = fopen(path, "rb");
	if (!fp) { rh_set_error("cannot open %s", path); return nullptr; }
	std::unique_ptr<rh_index_s> ix(new rh_index_s());
	char magic[2];
	uint32_t pars[7];
	auto fail = [&](const char *what) { fclose(fp); rh_set_error("%s: %s", path, what); return (rh_index*)nullptr; };
	auto remaining = [&]() -> uint64_t { const long at = ftell(fp); if (at < 0 || fseek(fp, 0, SEEK_END)) return 0; const long end = ftell(fp); fseek(fp, at, SEEK_SET); return end > at ? (uint64_t)(end - at) : 0; };
	const uint64_t file_bytes = remaining();
	if (!rd(fp, magic, 1, 2) || magic[0] != 'R' || magic[1] != 'I') return fail("not a RawHash2 index (magic)");
	if (!rd(fp, pars, 4, 7)) return fail("truncated header");
	ix->w = pars[0]; ix->e = pars[1]; ix->n = pars[2]; ix->q = pars[3]; ix->k = pars[4]; ix->flag = pars[6];
	const uint32_t n_seq = pars[5];
	if ((uint64_t)n_seq * 5 > file_bytes) return fail("implausible number of sequences");
	if (!rd(fp, &ix->diff, 4, 1) || !rd(fp, &ix->fine_min, 4, 1) || !rd(fp, &ix->fine_max, 4, 1) || !rd(fp, &ix->fine_range, 4, 1)) return fail("truncated header");
	unsigned char pore[32];
	if (!rd(fp, pore, 1, 32)) return fail("truncated pore header");
	memcpy(&ix->n_pore_vals, pore + 16, 4); memcpy(&ix->pore_k, pore + 20, 2);
	if (ix->n_pore_vals > (1u << 24)) return fail("implausible pore table size");
	ix->pore_vals.resize(ix->n_pore_vals); ix->pore_inds.resize((size_t)ix->n_pore_vals * 12);
	if (ix->n_pore_vals && (!rd(fp, ix->pore_vals.data(), 4, ix->n_pore_vals) || !rd(fp, ix->pore_inds.data(), 12, ix->n_pore_vals))) return fail("truncated pore tables");
	for (uint32_t i = 0; i < n_seq; ++i) {
		uint8_t l; uint32_t len;
		if (!rd(fp, &l, 1, 1)) return fail("truncated sequence table");
		std::string name(l, '\0');
		if (l && !rd(fp, &name[0], 1, l)) return fail("truncated sequence table");
		if (!rd(fp, &len, 4, 1)) return fail("truncated sequence table");
		ix->names.push_back(name); ix->lens.push_back(len);
		if (ix->flag & RH_I_STORE_SIG) {   // stored target signals (rindex.c:716-726): what DTW re-scoring aligns with
			uint32_t fl;
			if (!rd(fp, &fl, 4, 1) || (uint64_t)fl * 4 > remaining()) return fail("truncated stored signal");
			ix->sigF.emplace_back(fl);
			if (fl && !rd(fp, ix->sigF.back().data(), 4, fl)) return fail("truncated stored signal");
			if (!(ix->flag & RH_I_NO_REV_TARGET)) {
				if (!rd(fp, &fl, 4, 1) || (uint64_t)fl * 4 > remaining()) return fail("truncated stored signal");
				ix->sigR.emplace_back(fl);
				if (fl && !rd(fp, ix->sigR.back().data(), 4, fl)) return fail("truncated stored signal");
			}
		}
	}
	std::vector<Entry> ent;
	std::vector<uint64_t> kv;
	for (uint32_t b = 0; b < (1u << kBucketBits); ++b) {
		int32_t np; uint32_t size;
		if (!rd(fp, &np, 4, 1) || np < 0 || (uint64_t)np * 8 > remaining()) return fail("truncated bucket");
		const uint64_t base = ix->pos.size();
		ix->pos.resize(base + np);
		if (np && !rd(fp, ix->pos.data() + base, 8, np)) return fail("truncated bucket positions");
		if (!rd(fp, &size, 4, 1) || (uint64_t)size * 16 > remaining()) return fail("truncated bucket");
		kv.resize((size_t)size * 2);
		if (size && !rd(fp, kv.data(), 8, (size_t)size * 2)) return fail("truncated bucket keys");
		for (uint32_t j = 0; j < size; ++j) {
			const uint64_t key = kv[2 * j], val = kv[2 * j + 1];
			Entry e;
			e.hash = (uint32_t)(((key >> 1) << kBucketBits) | b);
			if (key & 1) { e.n = 1; e.val = val; }
			else { e.n = (uint32_t)val; e.val = base + (val >> 32); }
			ent.push_back(e);
		}
	}
	fclose(fp);
	std::sort(ent.begin(), ent.end(), [](const Entry &a, const Entry &b) { return a.hash < b.hash; });
	ix->key_hash.reserve(ent.size()); ix->key_n.reserve(ent.size()); ix->key_val.reserve(ent.size());
	for (const Entry &e : ent) { ix->key_hash.push_back(e.hash); ix->key_n.push_back(e.n); ix->key_val.push_back(e.val); }
	return ix.release();
}

extern "C" rh_index *rh_index_build(const char *fasta_path, const char *pore_model_path, const rh_idxopt_t *io, const char *out_ind, int n_threads)
{
	if (io->flag & RH_I_SIG_TARGET) { rh_set_error("signal-target (Rawsamble) index construction is not built yet"); return nullptr; }
	if (io->e < 1 || io->e > 16 || io->q < 1 || io->q * io->e > 64 || io->w < 0 || io->w > 255 || io->k < 1 || io->k > 12) { rh_set_error("unsupported index parameters e=%d q=%d w=%d k=%d", io->e, io->q, io->w, io->k); return nullptr; }
	std::unique_ptr<rh_index_s> ix(new rh_index_s());
	ix->w = io->w; ix->e = io->e; ix->n = io->n; ix->q = io->q; ix->k = io->k; ix->flag = io->flag;
	ix->diff = io->diff; ix->fine_min = io->fine_min; ix->fine_max = io->fine_max; ix->fine_range = io->fine_range;
	if (!load_model(pore_model_path, io->k, io->lev_col, ix->pore_vals)) return nullptr;
	ix->n_pore_vals = (uint32_t)ix->pore_vals.size(); ix->pore_k = (int16_t)io->k;
	make_pore_inds(ix->pore_vals, io->k, ix->pore_inds);
	std::vector<std::string> seqs;
	if (!read_fasta(fasta_path, ix->names, seqs)) return nullptr;
	for (const std::string &s : seqs) ix->lens.push_back((uint32_t)s.size());
	if (n_threads < 1) n_threads = 1;
	const rh_sketch_par sp = {io->e, io->w, io->q, io->k, io->diff, io->fine_min, io->fine_max, io->fine_range};
	// one task per (sequence, strand); results merged afterwards (order is irrelevant: everything is sorted below)
	const int n_strands = (io->flag & RH_I_NO_REV_TARGET) ? 1 : 2;
	const size_t n_tasks = seqs.size() * n_strands;
	std::vector<std::vector<HostSeed>> part(n_tasks);
	if (io->flag & RH_I_STORE_SIG) { ix->sigF.resize(seqs.size()); if (n_strands == 2) ix->sigR.resize(seqs.size()); }
	std::vector<std::thread> th;
	for (int t = 0; t < n_threads; ++t)
		th.emplace_back([&, t]() {
			std::vector<float> lv;
			for (size_t task = t; task < n_tasks; task += n_threads) {
				const size_t si = task / n_strands; const int strand = (int)(task % n_strands);
				seq_to_levels(seqs[si], ix->pore_vals, io->k, strand, lv);
				if (io->flag & RH_I_STORE_SIG) (strand ? ix->sigR : ix->sigF)[si] = lv;   // --store-sig (rindex.c:133-160)
				if (lv.empty()) continue;
				SeedSink sink{&part[task]};
				{ rh_sketch_store_local<256> st; rh_sketch_events<256>(lv.data(), (uint32_t)lv.size(), (uint32_t)si, strand, sp, sink, st); }
			}
		});
	for (auto &t : th) t.join();
	size_t total = 0;
	for (auto &p : part) total += p.size();
	std::vector<HostSeed> all;
	all.reserve(total);
	for (auto &p : part) { all.insert(all.end(), p.begin(), p.end()); std::vector<HostSeed>().swap(p); }
	finalize_keys(*ix, all, n_threads);
	if (out_ind && !write_ind(*ix, out_ind)) return nullptr;
	return ix.release();
}

extern "C" int rh_index_write(const rh_index *ix, const char *out_ind)
{
	if (ix->key_hash.empty() && ix->dev_n_keys) { rh_set_error("the keys of this index are on the device only: rh_index_download first"); return -1; }
	return write_ind(*ix, out_ind) ? 0 : -1;
}

extern "C" void rh_index_destroy(rh_index *ix) { delete ix; }
extern "C" uint32_t rh_index_n_seq(const rh_index *ix) { return (uint32_t)ix->names.size(); }
extern "C" const char *rh_index_seq_name(const rh_index *ix, uint32_t i) { return i < ix->names.size() ? ix->names[i].c_str() : nullptr; }
extern "C" uint32_t rh_index_seq_len(const rh_index *ix, uint32_t i) { return i < ix->lens.size() ? ix->lens[i] : 0; }
extern "C" uint64_t rh_index_n_keys(const rh_index *ix) { return ix->key_hash.empty() ? ix->dev_n_keys : ix->key_hash.size(); }
extern "C" uint64_t rh_index_n_positions(const rh_index *ix)
{
	if (ix->key_hash.empty()) return ix->dev_n_pos;
	uint64_t n = 0;
	for (uint32_t c : ix->key_n) n += c;
	return n;
}

extern "C" void rh_index_params(const rh_index *ix, rh_idxopt_t *o)
{
	*o = rh_idxopt_t{};
	o->b = kBucketBits; o->w = ix->w; o->e = ix->e; o->n = ix->n; o->q = ix->q; o->k = ix->k; o->flag = ix->flag; o->lev_col = 1;
	o->diff = ix->diff; o->fine_min = ix->fine_min; o->fine_max = ix->fine_max; o->fine_range = ix->fine_range;
}

extern "C" const uint64_t *rh_index_get(const rh_index *ix, uint64_t hashval, int *n)
{
	*n = 0;
	if (hashval >> 32) return nullptr;
	auto it = std::lower_bound(ix->key_hash.begin(), ix->key_hash.end(), (uint32_t)hashval);
	if (it == ix->key_hash.end() || *it != (uint32_t)hashval) return nullptr;
	const size_t i = it - ix->key_hash.begin();
	*n = (int)ix->key_n[i];
	return ix->key_n[i] == 1 ? &ix->key_val[i] : &ix->pos[ix->key_val[i]];
}

// mid_occ = (value at rank floor((1-f)*n_keys) of the per-key occurrence counts) + 1, clamped to [min_mid_occ, max_mid_occ]
extern "C" void rh_mapopt_update(rh_mapopt_t *mo, const rh_index *ix)
{
	if (mo->mid_occ <= 0) {
		int32_t thres = INT32_MAX;
		if (mo->mid_occ_frac > 0. && ix->key_n.empty() && ix->dev_n_keys) {
			// keys resident on the device only: the same order statistic from the occupancy histogram
			const uint64_t kk = (uint32_t)((1. - mo->mid_occ_frac) * ix->dev_n_keys);
			uint64_t run = 0;
			for (size_t v = 0; v < ix->occ_hist.size(); ++v) { run += ix->occ_hist[v]; if (run > kk) { thres = (int32_t)v + 1; break; } }
		} else if (mo->mid_occ_frac > 0. && !ix->key_n.empty()) {
			std::vector<uint32_t> a(ix->key_n);
			const size_t kk = (uint32_t)((1. - mo->mid_occ_frac) * a.size());
			std::nth_element(a.begin(), a.begin() + kk, a.end());
			thres = (int32_t)a[kk] + 1;
		}
		mo->mid_occ = thres;
		if (mo->mid_occ < mo->min_mid_occ) mo->mid_occ = mo->min_mid_occ;
		if (mo->max_mid_occ > mo->min_mid_occ && mo->mid_occ > mo->max_mid_occ) mo->mid_occ = mo->max_mid_occ;
	}
	if (mo->bw_long < mo->bw) mo->bw_long = mo->bw;
}

int rh_index_make_table(const rh_index_s &ix, std::vector<rh_tslot> &slots)
{
	// target <= 50 % slot occupancy: a bucket (one 128-byte line) overflows into the next one only rarely
	int lg = 4;
	while (((uint64_t)RH_TB_SLOTS << lg) < ix.key_hash.size() * 2) ++lg;
	const uint64_t nb = 1ULL << lg;
	slots.assign(nb * RH_TB_SLOTS, rh_tslot{0, 0, 0});
	for (size_t i = 0; i < ix.key_hash.size(); ++i) {
		const uint32_t h = ix.key_hash[i];
		uint64_t b = ((uint32_t)(h * 0x9E3779B1u)) >> (32 - lg);
		for (;;) {
			rh_tslot *s = &slots[b * RH_TB_SLOTS];
			int j = 0;
			while (j < RH_TB_SLOTS && s[j].n) ++j;
			if (j < RH_TB_SLOTS) { s[j] = rh_tslot{h, ix.key_n[i], ix.key_val[i]}; break; }
			b = (b + 1) & (nb - 1);
		}
	}
	return lg;
}
