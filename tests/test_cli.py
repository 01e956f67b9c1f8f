"""The drop-in `classify` executable (krakenuniq_amd/bin/classify): flag handling and exit codes on CPU,
byte-identical outputs against the reference's golden files on the GPU."""
import gzip
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "krakenuniq_amd", "bin", "classify")
F1 = os.path.join(ROOT, "tests", "golden", "f1")
F11 = os.path.join(ROOT, "tests", "golden", "f11")
DB = ["-d", f"{F1}/database.kdb", "-i", f"{F1}/database.idx", "-a", f"{F1}/taxDB"]


def run(args, **kw):
    return subprocess.run([BIN] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def rows(text):
    return sorted(text.strip("\n").split("\n"))


def test_usage_and_exit_codes():
    assert os.path.exists(BIN), "build with make -C krakenuniq_amd/csrc"
    r = run([])
    assert r.returncode == 64 and b"Missing mandatory option -d" in r.stderr  # EX_USAGE (classify.cpp:1151-1154)
    assert run(["-h"]).returncode == 0
    for bad in (["-t", "0"], ["-m", "0"], ["-u", "-5"], ["-x", "12Q"]):
        r = run(["-d", "x", "-i", "y"] + bad)
        assert r.returncode == 64, bad
    r = run(["-d", "x", "-i", "y", "r.fq"])
    assert r.returncode == 1 and b"TaxDB argument is required" in r.stderr  # classify.cpp:221-222
    r = run(["-d", "/nonexistent.kdb", "-i", f"{F1}/database.idx", "-a", f"{F1}/taxDB", "r.fq"])
    assert r.returncode == 66  # EX_NOINPUT
    r = run(["-d", f"{F1}/taxDB", "-i", f"{F1}/database.idx", "-a", f"{F1}/taxDB", "r.fq"])
    assert r.returncode == 65  # EX_DATAERR: not a JFLISTDN file
    assert run(DB + ["-I", "/nonexistent/uid.map", "r.fq"]).returncode == 66  # UID map: EX_NOINPUT
    r = run(DB + ["-I", f"{F11}/uid_to_taxid.map", "-q", "r.fq"])
    assert r.returncode == 1 and b"Quick mode not available when mapping UIDs" in r.stderr  # classify.cpp:954-956
    r = run(DB + ["-d", f"{F1}/database.kdb", "-i", f"{F1}/database.idx", "-I", f"{F11}/uid_to_taxid.map", "r.fq"])
    assert r.returncode == 1 and b"Cannot use more than one database with UID mapping" in r.stderr  # classify.cpp:158-160
    assert run(DB + ["-d", "second.kdb", "r.fq"]).returncode == 64  # a -d without its -i


@pytest.mark.gpu
def test_outputs_match_reference(tmp_path):
    shutil_db = tmp_path / "db"
    shutil_db.mkdir()
    for fn in ("database.kdb", "database.idx", "taxDB"):  # private copy: the CLI (re)writes database.kdb.counts
        (shutil_db / fn).write_bytes(open(f"{F1}/{fn}", "rb").read())
    db = ["-d", f"{shutil_db}/database.kdb", "-i", f"{shutil_db}/database.idx", "-a", f"{shutil_db}/taxDB"]
    out, rep = tmp_path / "out.tsv", tmp_path / "report.tsv"
    rep.write_text("# header written by the wrapper\n")  # the report is opened in append mode (classify.cpp:286)
    r = run(db + ["-t", "4", "-M", "-p", "14", "-o", str(out), "-r", str(rep), f"{F1}/reads.fq"])
    assert r.returncode == 0, r.stderr.decode()
    assert out.read_bytes() == open(f"{F1}/out.tsv", "rb").read()
    assert (shutil_db / "database.kdb.counts").read_text() == open(f"{F1}/database.kdb.counts").read()
    text = rep.read_text()
    assert text.startswith("# header written by the wrapper\n%\treads\ttaxReads\tkmers\tdup\tcov\ttaxID\trank\ttaxName\n")
    got = text.strip().split("\n")[1:]  # column header + rows
    # row for row the reference's report, `kmers` / `dup` / `cov` included (HLL sparse-mode emulation)
    assert sorted(got) == rows(open(f"{F1}/report.tsv").read())
    err = r.stderr.decode()
    assert "1000 sequences (0.15 Mbp) processed in" in err and "sequences classified (74.10%)" in err
    # stdout default, quick mode, -c, -s, gz output, -o off
    r = run(db + ["-q", "-m", "2", f"{F1}/reads.fq"])
    assert r.stdout == open(f"{F1}/out_quick.tsv", "rb").read()
    assert run(db + ["-c", f"{F1}/reads.fq"]).stdout == open(f"{F1}/out_c.tsv", "rb").read()
    assert run(db + ["-s", f"{F1}/reads.fq"]).stdout == open(f"{F1}/out_s.tsv", "rb").read()
    # -p 0 (and below): the six-column report, the Kraken file as ever (classify.cpp:289,316-323; golden from _ref/classify -p 0)
    for p_arg in ("0", "-3"):
        out0, rep0 = tmp_path / "out_p0.tsv", tmp_path / "report_p0.tsv"
        if rep0.exists():
            rep0.unlink()
        r = run(db + ["-p", p_arg, "-o", str(out0), "-r", str(rep0), f"{F1}/reads.fq"])
        assert r.returncode == 0, r.stderr.decode()
        assert out0.read_bytes() == open(f"{F1}/out.tsv", "rb").read()
        assert rep0.read_text().startswith("%\treads\ttaxReads\ttaxID\trank\ttaxName\n")
        assert rows(rep0.read_text()) == rows(open(f"{F1}/report_p0.tsv").read())
    assert run(db + ["-p", "x", f"{F1}/reads.fq"]).returncode == 64
    # one batch at a time (KU_RLE_ONE_STEP: the one-step form of the batch call) == two batches in flight (the default)
    r = run(db + ["-o", str(out), "-r", str(tmp_path / "rep_one_step.tsv"), f"{F1}/reads.fq"], env=dict(os.environ, KU_RLE_ONE_STEP="1"))
    assert r.returncode == 0 and out.read_bytes() == open(f"{F1}/out.tsv", "rb").read()
    assert rows((tmp_path / "rep_one_step.tsv").read_text()) == rows(open(f"{F1}/report.tsv").read())
    gz = tmp_path / "out.tsv.gz"
    assert run(db + ["-o", str(gz), f"{F1}/reads.fq"]).returncode == 0
    assert gzip.open(gz).read() == open(f"{F1}/out.tsv", "rb").read()
    # (the .gz file is ONE deflate stream whose parts the formatting helpers deflate, ku_pgzout.h: many batches and parts, parts
    #  without lines, and zlib's own writer beside it)
    assert run(db + ["-t", "3", "-o", str(gz), f"{F1}/reads.fq"], env={**os.environ, "KU_BATCH_NT": "65536"}).returncode == 0
    assert gzip.open(gz).read() == open(f"{F1}/out.tsv", "rb").read()
    assert run(db + ["-t", "16", "-c", "-o", str(gz), f"{F1}/reads.fq"], env={**os.environ, "KU_BATCH_NT": "65536"}).returncode == 0
    assert gzip.open(gz).read() == open(f"{F1}/out_c.tsv", "rb").read()
    assert run(db + ["-o", str(gz), f"{F1}/reads.fq"], env={**os.environ, "KU_NO_PGZOUT": "1"}).returncode == 0
    assert gzip.open(gz).read() == open(f"{F1}/out.tsv", "rb").read()
    r = run(db + ["-o", "off", f"{F1}/reads.fq"])
    assert r.returncode == 0 and r.stdout == b""
    # -u 1000: every per-unit sketch stays sparse (the reference's report_u1000.tsv); small GPU batches on top
    r = run(db + ["-u", "1000", "-o", str(out), "-r", str(tmp_path / "r2.tsv"), f"{F1}/reads.fq"], env={**os.environ, "KU_BATCH_NT": "65536"})
    assert out.read_bytes() == open(f"{F1}/out_u1000.tsv", "rb").read()
    assert rows((tmp_path / "r2.tsv").read_text()) == rows(open(f"{F1}/report_u1000.tsv").read())
    # without the emulation: dense estimates, everything else identical
    r = run(db + ["-o", str(out), "-r", str(tmp_path / "r3.tsv"), f"{F1}/reads.fq"], env={**os.environ, "KU_NO_SPARSE": "1"})
    ref = {ln.split("\t")[6]: ln.split("\t") for ln in open(f"{F1}/report.tsv").read().strip().split("\n")}
    for ln in (tmp_path / "r3.tsv").read_text().strip().split("\n"):
        f = ln.split("\t")
        assert f[:3] == ref[f[6]][:3] and f[6:] == ref[f[6]][6:]
        if f[3] != "kmers":
            assert abs(int(f[3]) - int(ref[f[6]][3])) <= max(2, 3 * 0.01625 * int(ref[f[6]][3]))


@pytest.mark.gpu
def test_fasta_paired_edge_and_read_files(tmp_path):
    g = os.path.join(ROOT, "tests", "golden")
    r = run(DB + [f"{g}/f2/edge.fa"])
    assert r.returncode == 0 and r.stdout == open(f"{g}/f2/out.tsv", "rb").read()
    r = run(DB + [f"{g}/f4/merged.fa"])
    assert r.stdout == open(f"{g}/f4/out.tsv", "rb").read()
    # gz input and two input files in one run
    gzp = tmp_path / "edge.fa.gz"
    with gzip.open(gzp, "wb") as f:
        f.write(open(f"{g}/f2/edge.fa", "rb").read())
    r = run(DB + [str(gzp), f"{g}/f4/merged.fa"])
    assert r.stdout == open(f"{g}/f2/out.tsv", "rb").read() + open(f"{g}/f4/out.tsv", "rb").read()
    # -C / -U: classified and unclassified reads as FASTQ records
    c, u = tmp_path / "c.fq", tmp_path / "u.fq"
    r = run(DB + ["-o", "off", "-C", str(c), "-U", str(u), f"{F1}/reads.fq"])
    assert r.returncode == 0
    calls = {ln.split("\t")[1]: ln[0] for ln in open(f"{F1}/out.tsv").read().strip().split("\n")}
    src = open(f"{F1}/reads.fq").read().split("\n")
    want_c = "".join("\n".join(src[i:i + 4]) + "\n" for i in range(0, len(src) - 1, 4) if calls[src[i][1:]] == "C")
    want_u = "".join("\n".join(src[i:i + 4]) + "\n" for i in range(0, len(src) - 1, 4) if calls[src[i][1:]] == "U")
    assert c.read_text() == want_c and u.read_text() == want_u


@pytest.mark.gpu
def test_end_of_input_rules_equal_the_reference(tmp_path):
    """tests/golden/f12 (make_golden.py f12): how a file's input ends -- a FASTA header as the last line without a line end, a
    work unit without nucleotides (src/classify.cpp:510-523, decided per WORK UNIT, not per GPU batch), damaged FASTQ files whose
    regions the cutter gets wrong -- byte for byte the reference's `classify -s` output, with the parser team, with one reader
    thread, from .gz, and with GPU batches so small that the files span several"""
    import json
    d = os.path.join(ROOT, "tests", "golden", "f12")
    for case in json.load(open(f"{d}/cases.json")):
        want = open(f"{d}/{case['output']}", "rb").read()
        path = f"{d}/{case['input']}"
        gz = tmp_path / (case["input"] + ".gz")
        with gzip.open(gz, "wb") as f:
            f.write(open(path, "rb").read())
        for threads, src, env in (("4", path, {}), ("1", path, {}), ("4", path, {"KU_BATCH_NT": "65536", "KU_REGION_RAMP": "0"}),
                                  ("4", str(gz), {}), ("1", str(gz), {})):
            r = run(DB + ["-s", "-t", threads] + case["flags"] + [src], env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr.decode()
            assert r.stdout == want, (case, threads, src, env)
    # a stretch of empty records longer than a GPU batch in the middle of a file: batches without a nucleotide travel through the
    # device like any other (they share a work unit with the reads that follow), at the end of a file they are a unit of their own
    head = open(f"{d}/empty_inside.fa", "rb").read().split(b">e1\n")[0]
    lines = open(f"{d}/empty_inside.fa.u1500.out.tsv", "rb").read().split(b"\n")[:-1]
    tail = b"".join(b">" + ln.split(b"\t")[1] + b"\n" + ln.split(b"\t")[5] + b"\n" for ln in lines[23:])
    empties = b"".join(b">x%d\n" % j for j in range(6000))
    want_empties = b"".join(b"U\tx%d\t0\t0\t0:0\t\n" % j for j in range(6000))
    for name, data, want in (("inside.fa", head + empties + tail, b"\n".join(lines[:20]) + b"\n" + want_empties + b"\n".join(lines[23:]) + b"\n"),
                             ("behind.fa", head + empties, b"\n".join(lines[:20]) + b"\n")):
        f = tmp_path / name
        f.write_bytes(data)
        for threads, env in (("4", {"KU_BATCH_NT": "65536", "KU_REGION_RAMP": "0"}), ("1", {"KU_BATCH_NT": "65536"}), ("4", {})):
            r = run(DB + ["-s", "-u", "1500", "-t", threads, str(f)], env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr.decode()
            assert r.stdout == want, (name, threads, env)
            # and with a report: the sparse-sketch emulation's unit plan sees the same reads
            rep = tmp_path / "rep.tsv"
            if rep.exists():
                rep.unlink()
            r = run(DB + ["-s", "-u", "1500", "-t", threads, "-r", str(rep), str(f)], env=dict(os.environ, **env))
            assert r.returncode == 0 and r.stdout == want, r.stderr.decode()


@pytest.mark.gpu
def test_crlf_inputs():
    """tests/golden/f10: CRLF FASTQ and one-line-per-sequence CRLF FASTA are byte-identical with the reference (the '\\r'
    closing a sequence is one more, ambiguous, base).  Multi-line CRLF FASTA is the one documented deviation: the
    reference's scanner loses one k-mer per line break there (src/krakenutil.cpp:266-270; crlf_multiline.out.tsv keeps
    its output); here the line breaks are removed, said once on stderr, so the reads classify like the one-line file --
    unless KU_CRLF_REFERENCE=1 asks for the reference's lines"""
    g = os.path.join(ROOT, "tests", "golden", "f10")
    for name in ("crlf.fq", "crlf_oneline.fa"):
        r = run(DB + [f"{g}/{name}"])
        assert r.returncode == 0 and b"CRLF" not in r.stderr
        assert r.stdout == open(f"{g}/{name.rsplit('.', 1)[0]}.out.tsv", "rb").read(), name
    r = run(DB + [f"{g}/crlf_multiline.fa"])
    assert r.returncode == 0 and r.stderr.count(b"multi-line FASTA with CRLF") == 1
    assert r.stdout == open(f"{g}/crlf_oneline.out.tsv", "rb").read()
    ref = open(f"{g}/crlf_multiline.out.tsv").read().splitlines()
    assert [ln.split("\t")[:3] for ln in r.stdout.decode().splitlines()] == [ln.split("\t")[:3] for ln in ref]
    # KU_CRLF_REFERENCE=1 (round 5): the reference's lines for such reads -- the k-mer behind every line break taken out of the hit
    # list, the carriage returns counted in the length column -- byte for byte its output file; also through several threads'
    # parts, the parser team and a .gz file
    for extra, name in ((["-t", "1"], "crlf_multiline.fa"), (["-t", "4"], "crlf_multiline.fa")):
        r = run(DB + extra + [f"{g}/{name}"], env=dict(os.environ, KU_CRLF_REFERENCE="1"))
        assert r.returncode == 0 and b"multi-line FASTA with CRLF" not in r.stderr
        assert r.stdout == open(f"{g}/crlf_multiline.out.tsv", "rb").read(), extra
    for name in ("crlf.fq", "crlf_oneline.fa"):  # (nothing changes for the files that agreed already)
        r = run(DB + [f"{g}/{name}"], env=dict(os.environ, KU_CRLF_REFERENCE="1"))
        assert r.stdout == open(f"{g}/{name.rsplit('.', 1)[0]}.out.tsv", "rb").read(), name


@pytest.mark.gpu
def test_hierarchical_multi_db_run(tmp_path):
    """-d A -i A.idx -d B -i B.idx (classify.cpp:163-177,928-936): both database orders against the reference's outputs"""
    g = os.path.join(ROOT, "tests", "golden")
    dirs = {}
    for name in ("f1", "f8"):  # private copies: the CLI writes database.kdb.counts next to each database
        d = tmp_path / name
        d.mkdir()
        for fn in ("database.kdb", "database.idx"):
            (d / fn).write_bytes(open(f"{g}/{name}/{fn}", "rb").read())
        dirs[name] = d
    a = ["-d", f"{dirs['f1']}/database.kdb", "-i", f"{dirs['f1']}/database.idx"]
    b = ["-d", f"{dirs['f8']}/database.kdb", "-i", f"{dirs['f8']}/database.idx"]
    common = ["-a", f"{F1}/taxDB", f"{g}/f8/reads.fq"]
    for first, second, tag in ((a, b, ""), (b, a, "_swapped")):
        out, rep = tmp_path / f"out{tag}.tsv", tmp_path / f"rep{tag}.tsv"
        r = run(first + second + ["-o", str(out), "-r", str(rep)] + common)
        assert r.returncode == 0, r.stderr.decode()
        assert out.read_bytes() == open(f"{g}/f8/out{tag}.tsv", "rb").read()
        assert rows(rep.read_text()) == rows(open(f"{g}/f8/report{tag}.tsv").read())  # row for row, `kmers` included
    for name in ("f1", "f8"):
        assert (dirs[name] / "database.kdb.counts").read_text() == open(f"{g}/{name}/database.kdb.counts").read()
    assert run(a + b + ["-q", "-m", "2"] + common).stdout == open(f"{g}/f8/out_quick.tsv", "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("size", ["70K", "30K"])
def test_out_of_core_chunked_run(tmp_path, size):
    """-x SIZE: database streamed through HBM in chunks (src/classify.cpp:566-791) -> the reference's -x outputs;
    the counts file is summed up over the chunks"""
    d = tmp_path / "db"
    d.mkdir()
    for fn in ("database.kdb", "database.idx", "taxDB"):
        (d / fn).write_bytes(open(f"{F1}/{fn}", "rb").read())
    db = ["-d", f"{d}/database.kdb", "-i", f"{d}/database.idx", "-a", f"{d}/taxDB"]
    out, rep = tmp_path / "out.tsv", tmp_path / "rep.tsv"
    r = run(db + ["-x", size, "-t", "2", "-u", "1", "-o", str(out), "-r", str(rep), f"{F1}/reads.fq"])
    assert r.returncode == 0, r.stderr.decode()
    assert b"Streaming the database through the GPU in" in r.stderr
    assert out.read_bytes() == open(f"{F1}/out_chunk.tsv", "rb").read()
    assert (d / "database.kdb.counts").read_text() == open(f"{F1}/database.kdb.counts").read()
    assert rows(rep.read_text()) == rows(open(f"{F1}/report_chunk.tsv").read())  # row for row, `kmers` included
    # bounded memory: a device budget of 100 kB holds a fraction of the reads at a time, so the chunks cycle several
    # times (one pass over all chunks per super-batch); outputs and report stay those of the reference
    env = {**os.environ, "KU_SUPERBATCH_BYTES": "100000", "KU_BATCH_NT": "65536"}
    (d / "database.kdb.counts").unlink()
    r2 = run(db + ["-x", size, "-t", "2", "-o", str(out), "-r", str(tmp_path / "rep2.tsv"), f"{F1}/reads.fq"], env=env)
    assert r2.returncode == 0, r2.stderr.decode()
    assert b"passes over the" in r2.stderr
    assert out.read_bytes() == open(f"{F1}/out_chunk.tsv", "rb").read()
    assert rows((tmp_path / "rep2.tsv").read_text()) == rows(open(f"{F1}/report_chunk.tsv").read())
    assert (d / "database.kdb.counts").read_text() == open(f"{F1}/database.kdb.counts").read()
    # quick mode inside a chunked run has the reference's own semantics (classify.cpp:686-737)
    (d / "database.kdb.counts").unlink()
    r3 = run(db + ["-x", size, "-t", "2", "-q", "-m", "2", "-o", str(out), "-r", str(tmp_path / "rep3.tsv"), f"{F1}/reads.fq"])
    assert r3.returncode == 0, r3.stderr.decode()
    assert out.read_bytes() == open(f"{F1}/out_chunk_quick.tsv", "rb").read()
    assert rows((tmp_path / "rep3.tsv").read_text()) == rows(open(f"{F1}/report_chunk_quick.tsv").read())
    # identical to the run with everything resident, FASTA + second file included
    g = os.path.join(ROOT, "tests", "golden")
    r1 = run(db + ["-x", size, f"{g}/f2/edge.fa", f"{g}/f4/merged.fa"])
    assert r1.stdout == open(f"{g}/f2/out.tsv", "rb").read() + open(f"{g}/f4/out.tsv", "rb").read()
    # a budget smaller than the largest bin is a usage error, as in the reference
    assert run(db + ["-x", "1K", f"{F1}/reads.fq"]).returncode != 0


@pytest.mark.gpu
def test_a_preload_size_that_holds_the_whole_database_is_still_the_chunk_mode(tmp_path):
    """-x SIZE with a plan of ONE chunk: the reference runs its chunk mode all the same (src/classify.cpp:196-198,251-252) --
    quick mode calls the taxon of the read's last unambiguous k-mer, -u is ignored (one work unit for the run); found by
    tests/fuzz_cli.py.  The goldens are the reference's -x 70K outputs, which its -x 10M run reproduces byte for byte."""
    d = tmp_path / "db"
    d.mkdir()
    for fn in ("database.kdb", "database.idx", "taxDB"):
        (d / fn).write_bytes(open(f"{F1}/{fn}", "rb").read())
    db = ["-d", f"{d}/database.kdb", "-i", f"{d}/database.idx", "-a", f"{d}/taxDB"]
    out, rep = tmp_path / "out.tsv", tmp_path / "rep.tsv"
    r = run(db + ["-x", "10M", "-t", "3", "-q", "-m", "2", "-o", str(out), "-r", str(rep), f"{F1}/reads.fq"])
    assert r.returncode == 0, r.stderr.decode()
    assert out.read_bytes() == open(f"{F1}/out_chunk_quick.tsv", "rb").read()
    assert out.read_bytes() != open(f"{F1}/out_quick.tsv", "rb").read()  # (the plain run's quick mode calls differently)
    assert rows(rep.read_text()) == rows(open(f"{F1}/report_chunk_quick.tsv").read())
    # without -q everything stays resident (no streaming), with the chunk mode's accounting: -u does not apply
    r = run(db + ["-x", "10M", "-u", "1000", "-o", str(out), "-r", str(tmp_path / "rep2.tsv"), f"{F1}/reads.fq"])
    assert r.returncode == 0 and b"Streaming the database" not in r.stderr
    assert out.read_bytes() == open(f"{F1}/out_chunk.tsv", "rb").read()
    assert rows((tmp_path / "rep2.tsv").read_text()) == rows(open(f"{F1}/report_chunk.tsv").read())
    assert rows(open(f"{F1}/report_chunk.tsv").read()) != rows(open(f"{F1}/report_u1000.tsv").read())


@pytest.mark.gpu
@pytest.mark.parametrize("devices,size", [("0,0", "30K"), ("0,0,0", "30K"), ("0,0,0,0,0,0,0", "70K")])
def test_out_of_core_run_on_several_gpus(tmp_path, devices, size):
    """KU_DEVICES with -x: the chunks are dealt out among the GPUs (chunk c on GPU c mod N; more GPUs than chunks: the rest
    stay idle), every GPU streams its share over its own copies of the resident batches, the slots are folded into the
    first GPU's batches before the finish, the per-taxon state at the end -- the reference's -x files, the counts file summed
    over all GPUs' chunks, several super-batches, quick mode"""
    d = tmp_path / "db"
    d.mkdir()
    for fn in ("database.kdb", "database.idx", "taxDB"):
        (d / fn).write_bytes(open(f"{F1}/{fn}", "rb").read())
    db = ["-d", f"{d}/database.kdb", "-i", f"{d}/database.idx", "-a", f"{d}/taxDB"]
    out, rep = tmp_path / "out.tsv", tmp_path / "rep.tsv"
    env = {**os.environ, "KU_DEVICES": devices}
    r = run(db + ["-x", size, "-t", "2", "-o", str(out), "-r", str(rep), f"{F1}/reads.fq"], env=env)
    assert r.returncode == 0, r.stderr.decode()
    assert b"the database chunks of the out-of-core run are dealt out among them" in r.stderr
    assert out.read_bytes() == open(f"{F1}/out_chunk.tsv", "rb").read()
    assert (d / "database.kdb.counts").read_text() == open(f"{F1}/database.kdb.counts").read()
    assert rows(rep.read_text()) == rows(open(f"{F1}/report_chunk.tsv").read())
    env2 = {**env, "KU_SUPERBATCH_BYTES": "100000", "KU_BATCH_NT": "65536"}
    (d / "database.kdb.counts").unlink()
    r2 = run(db + ["-x", size, "-t", "2", "-o", str(out), "-r", str(tmp_path / "rep2.tsv"), f"{F1}/reads.fq"], env=env2)
    assert r2.returncode == 0, r2.stderr.decode()
    assert b"passes over the" in r2.stderr
    assert out.read_bytes() == open(f"{F1}/out_chunk.tsv", "rb").read()
    assert rows((tmp_path / "rep2.tsv").read_text()) == rows(open(f"{F1}/report_chunk.tsv").read())
    assert (d / "database.kdb.counts").read_text() == open(f"{F1}/database.kdb.counts").read()
    r3 = run(db + ["-x", size, "-t", "2", "-q", "-m", "2", "-o", str(out), "-r", str(tmp_path / "rep3.tsv"), f"{F1}/reads.fq"], env=env)
    assert r3.returncode == 0, r3.stderr.decode()
    assert out.read_bytes() == open(f"{F1}/out_chunk_quick.tsv", "rb").read()
    assert rows((tmp_path / "rep3.tsv").read_text()) == rows(open(f"{F1}/report_chunk_quick.tsv").read())
    # a preload size that takes the whole database: no chunks, the group runs sharded as without -x
    r4 = run(db + ["-x", "10M", "-t", "2", "-o", str(out), f"{F1}/reads.fq"], env=env)
    assert r4.returncode == 0 and b"database sharded by minimizer range" in r4.stderr
    assert out.read_bytes() == open(f"{F1}/out.tsv", "rb").read()


@pytest.mark.gpu
def test_mate_pairs_merged_on_the_fly(tmp_path):
    """-P r_1.fq r_2.fq == read_merger.pl | classify (scripts/krakenuniq:230-238): the reference's f4 outputs"""
    g = os.path.join(ROOT, "tests", "golden")
    r = run(DB + ["-P", f"{g}/f4/r_1.fq", f"{g}/f4/r_2.fq"])
    assert r.returncode == 0 and r.stdout == open(f"{g}/f4/out.tsv", "rb").read()
    # classified / unclassified records are the merged FASTA records classify would have seen
    c, u = tmp_path / "c.fa", tmp_path / "u.fa"
    assert run(DB + ["-P", "-o", "off", "-C", str(c), "-U", str(u), f"{g}/f4/r_1.fq", f"{g}/f4/r_2.fq"]).returncode == 0
    calls = {ln.split("\t")[1]: ln[0] for ln in open(f"{g}/f4/out.tsv").read().strip().split("\n")}
    merged = open(f"{g}/f4/merged.fa").read().split("\n")
    recs = [(merged[i][1:], merged[i + 1]) for i in range(0, len(merged) - 1, 2)]
    assert c.read_text() == "".join(f">{i}\n{s}\n" for i, s in recs if calls[i] == "C")
    assert u.read_text() == "".join(f">{i}\n{s}\n" for i, s in recs if calls[i] == "U")
    assert run(DB + ["-P", f"{g}/f4/r_1.fq"]).returncode == 64


@pytest.mark.gpu
def test_classify_exact_binary(tmp_path):
    """bin/classifyExact: same Kraken output, report with exact distinct k-mer counts == the reference's classifyExact"""
    exact = os.path.join(ROOT, "krakenuniq_amd", "bin", "classifyExact")
    d = tmp_path / "db"
    d.mkdir()
    for fn in ("database.kdb", "database.idx", "taxDB", "database.kdb.counts"):
        (d / fn).write_bytes(open(f"{F1}/{fn}", "rb").read())
    out, rep = tmp_path / "out.tsv", tmp_path / "rep.tsv"
    r = subprocess.run([exact, "-d", f"{d}/database.kdb", "-i", f"{d}/database.idx", "-a", f"{d}/taxDB", "-o", str(out),
                        "-r", str(rep), f"{F1}/reads.fq"], stderr=subprocess.PIPE, env=dict(os.environ, KU_EXACT_LOG2="18"))
    assert r.returncode == 0, r.stderr.decode()
    assert out.read_bytes() == open(f"{F1}/out.tsv", "rb").read()
    assert rows(rep.read_text()) == rows(open(f"{F1}/report_exact.tsv").read())
    # classifyExact is the reference's classify with every flag (classify.cpp:46-53): quick mode counts the scanned prefix
    # of each read, a chunked run every k-mer (golden: oracle/_ref/classifyExact -q -m 2 / -x 70K -t 2)
    for extra, want_out, want_rep in ((["-q", "-m", "2"], "out_quick.tsv", "report_exact_quick.tsv"),
                                      (["-x", "70K", "-t", "2"], "out_chunk.tsv", "report_exact_chunk.tsv")):
        rep.unlink()  # (the report is appended to what the wrapper put there)
        r = subprocess.run([exact, "-d", f"{d}/database.kdb", "-i", f"{d}/database.idx", "-a", f"{d}/taxDB", "-o", str(out),
                            "-r", str(rep), *extra, f"{F1}/reads.fq"], stderr=subprocess.PIPE, env=dict(os.environ, KU_EXACT_LOG2="18"))
        assert r.returncode == 0, r.stderr.decode()
        assert out.read_bytes() == open(f"{F1}/{want_out}", "rb").read(), extra
        assert rows(rep.read_text()) == rows(open(f"{F1}/{want_rep}").read()), extra


@pytest.mark.gpu
def test_report_says_when_the_sparse_emulation_gave_up(tmp_path):
    """no device memory for the run-wide (slot, encoding) set (test hook: a ceiling of 2^11 cells): the run goes on, the
    Kraken file is the reference's, stderr AND the report file say that kmers / dup / cov are dense estimates"""
    out, rep = tmp_path / "out.tsv", tmp_path / "rep.tsv"
    r = subprocess.run([BIN] + DB + ["-u", "1000", "-o", str(out), "-r", str(rep), f"{F1}/reads.fq"], stderr=subprocess.PIPE,
                       env=dict(os.environ, KU_SPARSE_MAX_LOG2="11", KU_SPARSE_LOG2="10"))
    assert r.returncode == 0, r.stderr.decode()
    assert b"dense-register estimates" in r.stderr
    assert out.read_bytes() == open(f"{F1}/out_u1000.tsv", "rb").read()
    text = rep.read_text()
    assert text.startswith("# NOTE: kmers / dup / cov are dense HyperLogLog estimates")
    body = [l for l in text.split("\n") if l and not l.startswith("#")]
    want = [l for l in open(f"{F1}/report_u1000.tsv").read().split("\n") if l and not l.startswith("#")]
    assert len(body) == len(want) and [l.split("\t")[1:3] for l in body] == [l.split("\t")[1:3] for l in want]  # reads / taxReads


@pytest.mark.gpu
def test_pipe_inputs(tmp_path):
    """classify <(cat reads.fq) and a gzip stream through a pipe (the wrapper and build_db.sh hand over such paths)"""
    cmd = f"{BIN} {' '.join(DB)} -t 4 <(cat {F1}/reads.fq) <(gzip -c {F1}/reads.fq)"
    r = subprocess.run(["bash", "-c", cmd], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    want = open(f"{F1}/out.tsv", "rb").read()
    assert r.stdout == want + want


@pytest.mark.gpu
@pytest.mark.parametrize("unit,rep", [(None, "report_uid.tsv"), ("1000", "report_uid_u1000.tsv")])
def test_uid_mapping_matches_reference(tmp_path, unit, rep):
    """classify -I on the UID database the reference's set_lcas -I built (tests/golden/f11): the Kraken file (per-k-mer
    codes are UIDs, calls come from resolve_uids3) and the report of the reference's classify -I, row for row"""
    d = tmp_path / "db"
    d.mkdir()
    for fn, src in (("uid_database.kdb", F11), ("uid_database.kdb.counts", F11), ("uid_to_taxid.map", F11), ("database.idx", F1), ("taxDB", F1)):
        (d / fn).write_bytes(open(f"{src}/{fn}", "rb").read())
    out, report = tmp_path / "out.tsv", tmp_path / "report.tsv"
    args = ["-d", f"{d}/uid_database.kdb", "-i", f"{d}/database.idx", "-a", f"{d}/taxDB", "-I", f"{d}/uid_to_taxid.map", "-t", "4",
            "-o", str(out), "-r", str(report)] + (["-u", unit] if unit else []) + [f"{F1}/reads.fq"]
    r = run(args, env=dict(os.environ, KU_BATCH_NT="65536"))  # several device batches
    assert r.returncode == 0, r.stderr.decode()
    assert out.read_bytes() == open(f"{F11}/out_uid.tsv", "rb").read()
    assert rows(report.read_text()) == rows(open(f"{F11}/{rep}").read())
    assert b"Reading UID mapping file" in r.stderr


@pytest.mark.gpu
def test_gz_input_is_parsed_in_regions_like_the_plain_file(tmp_path):
    """a regular .gz file (one gzip stream, or BGZF) or .bz2 file is inflated by a team into text that the parser team cuts into regions
    while it arrives (ku_seqio.h GrowingText / RegionCutter, ku_pgzip.h): Kraken output and report identical to the plain
    file's and to the sequential reader's (KU_NO_GZ_REGIONS=1), with small batches / spans / look-ahead (many regions and
    rounds), -C/-U records included; a truncated file is a data error of the run"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_seqio import bgzf
    src = open(f"{F1}/reads.fq", "rb").read().split(b"\n")
    recs = [src[i:i + 4] for i in range(0, len(src) - 1, 4)]
    text = b"".join(b"@" + r[0][1:] + b"_%d extra words\n" % rep + r[1] + b"\n+\n" + r[3] + b"\n" for rep in range(30) for r in recs)
    plain = tmp_path / "big.fq"
    plain.write_bytes(text)
    z = tmp_path / "big.fq.gz"
    z.write_bytes(gzip.compress(text, 6))
    b = tmp_path / "big.bgzf.fq.gz"
    b.write_bytes(bgzf(text, 65280))
    import bz2
    bz = tmp_path / "big.fq.bz2"
    bz.write_bytes(bz2.compress(text[:4000000], 2) + bz2.compress(text[4000000:], 9))  # (two streams, blocks of 200 and 900 kB)

    def go(path, tag, **env):
        out, rep, c = tmp_path / f"{tag}.tsv", tmp_path / f"{tag}.rep", tmp_path / f"{tag}.c.fq"
        r = run(DB + ["-t", "8", "-o", str(out), "-r", str(rep), "-C", str(c), str(path)], env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()
        return out.read_bytes(), rows(rep.read_text()), c.read_bytes()
    want = go(plain, "plain")
    assert want[0].count(b"\n") == 30 * len(recs)
    small = {"KU_BATCH_NT": "1000000", "KU_PGZIP_SPAN_KB": "64", "KU_TEXT_AHEAD_MB": "1"}
    want_small = go(plain, "plain_small", **small)  # (the regions are cut by the text alone: the same ones, plain or .gz)
    assert (want_small[0], want_small[2]) == (want[0], want[2])
    for tag, path in (("gz", z), ("bgzf", b), ("bz2", bz)):
        assert go(path, tag) == want, tag
        assert go(path, tag + "_seq", KU_NO_GZ_REGIONS="1") == want, tag
        assert go(path, tag + "_small", **small) == want_small, tag
    blob = z.read_bytes()
    cut = tmp_path / "cut.fq.gz"
    cut.write_bytes(blob[:len(blob) // 2])
    r = run(DB + ["-t", "8", "-o", "off", str(cut)])
    assert r.returncode == 65 and b"gzip" in r.stderr
    cut2 = tmp_path / "cut.fq.bz2"
    cut2.write_bytes(bz.read_bytes()[:bz.stat().st_size // 2])
    r = run(DB + ["-t", "8", "-o", "off", str(cut2)])
    assert r.returncode == 65 and b"bzip2" in r.stderr
    # mate pairs from compressed files: the sequential readers (each with its team) == the plain files
    half = len(recs) * 15
    t1 = b"".join(b"@" + r[0][1:] + b"/1\n" + r[1] + b"\n+\n" + r[3] + b"\n" for rep in range(15) for r in recs)
    t2 = b"".join(b"@" + r[0][1:] + b"/2\n" + r[1][::-1] + b"\n+\n" + r[3] + b"\n" for rep in range(15) for r in recs)
    files = {}
    for nm, t in (("m1", t1), ("m2", t2)):
        (tmp_path / f"{nm}.fq").write_bytes(t)
        (tmp_path / f"{nm}.fq.gz").write_bytes(gzip.compress(t, 1))
        (tmp_path / f"{nm}.fq.bz2").write_bytes(bz2.compress(t, 1))
    outs = []
    for ext in ("fq", "fq.gz", "fq.bz2"):
        r = run(DB + ["-t", "8", "-P", str(tmp_path / f"m1.{ext}"), str(tmp_path / f"m2.{ext}")])
        assert r.returncode == 0, r.stderr.decode()
        outs.append(r.stdout)
    assert outs[0].count(b"\n") == half and outs[1] == outs[0] and outs[2] == outs[0]
