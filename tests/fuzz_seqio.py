"""Differential fuzzer of the input stage: bin/seqio_dump (sequential, and -j N region-parallel, plain and .gz) against the
REFERENCE reader as `oracle/_ref/classify -s` shows it (src/seqreader.cpp:26-133 + the work-unit rule of
src/classify.cpp:510-523).  Test infrastructure: runs in the build container, where the compiled reference exists
(tests/test_seqio.py::test_fuzz_against_reference runs a bounded number of cases; run this file for more).

    python tests/fuzz_seqio.py [n_cases] [seed]

A case = a small FASTA or FASTQ file made of well-formed records plus a few edits of the kinds that break real files: deleted /
duplicated / blanked lines, a missing final line end, a header as the last line, runs of empty records, quality lines that start
with '@' or '+', CRLF line ends (FASTQ only: CRLF inside multi-line FASTA is a documented difference).
"""
import gzip
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = os.path.join(ROOT, "krakenuniq_amd", "bin", "seqio_dump")
REF = os.path.join(ROOT, "oracle", "_ref", "classify")
F1 = os.path.join(ROOT, "tests", "golden", "f1")


def make_case(rng):
    """-> (bytes of the file, is_fastq, work unit size)"""
    fastq = rng.random() < 0.6
    n = rng.choice([1, 2, 3, 5, 8, 13, 40])
    hostile = rng.random() < 0.2  # lines that look like the start of another kind of line, often
    lines = []
    for i in range(n):
        L = rng.choice([0, 0, 1, 5, 30, 31, 40, 70, 150])
        seq = "".join(rng.choice("ACGTACGTACGTN") for _ in range(L))
        if L and hostile and rng.random() < 0.7:
            seq = rng.choice("+@>") + seq[1:]
        hdr = f"r{i}" + rng.choice(["", " desc", "\tx y", "/1"])
        if fastq:
            q = "".join(rng.choice("IIIIFF#@+5") for _ in range(L))
            if L and rng.random() < (0.8 if hostile else 0.15):
                q = rng.choice("@+") + q[1:]
            lines += ["@" + hdr, seq, rng.choice(["+", "+", "+" + hdr]), q]
        else:
            lines.append(">" + hdr)
            if L and rng.random() < 0.3:  # several lines
                w = rng.choice([7, 20, 60])
                lines += [seq[j:j + w] for j in range(0, L, w)]
            elif L or rng.random() < 0.5:
                lines.append(seq)
    # edits
    for _ in range(rng.choice([0, 0, 1, 1, 2, 3])):
        if not lines:
            break
        k = rng.randrange(len(lines))
        what = rng.choice(["del", "dup", "blank", "swap", "hdr_tail", "empties", "plus", "at"])
        if what == "del":
            del lines[k]
        elif what == "dup":
            lines.insert(k, lines[k])
        elif what == "blank":
            lines[k] = ""
        elif what == "swap" and k + 1 < len(lines):
            lines[k], lines[k + 1] = lines[k + 1], lines[k]
        elif what == "hdr_tail":
            lines.append(("@" if fastq else ">") + "tail")
        elif what == "empties":
            m = rng.choice([1, 2, 5])
            at = rng.choice([k, len(lines)])
            add = []
            for j in range(m):
                add += (["@e%d" % j, "", "+", ""] if fastq else [">e%d" % j] + ([""] if rng.random() < 0.3 else []))
            lines[at:at] = add
        elif what == "plus":
            lines[k] = "+" + lines[k]
        elif what == "at":
            lines[k] = "@" + lines[k]
    nl = "\r\n" if (fastq and rng.random() < 0.08) else "\n"
    text = nl.join(lines)
    if lines and rng.random() < 0.75:
        text += nl
    unit = rng.choice([1, 30, 100, 150, 500, 500000])
    return text.encode(), fastq, unit


def reference_records(path, unit):
    """(id, sequence) pairs as the reference's classify prints them with -s (columns 2 and 6)"""
    r = subprocess.run([REF, "-d", f"{F1}/database.kdb", "-i", f"{F1}/database.idx", "-a", f"{F1}/taxDB", "-s", "-t", "1",
                        "-u", str(unit), path], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    out = []
    for ln in r.stdout.split(b"\n")[:-1]:
        c = ln.split(b"\t", 5)
        assert len(c) == 6, ln
        out.append((c[1], c[5]))
    return out


def dump_records(args):
    r = subprocess.run([DUMP] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, (args, r.stderr.decode())
    return [tuple(ln.split(b"\t", 1)) for ln in r.stdout.split(b"\n")[:-1]]


def run_case(data, fastq, unit, d, modes=("seq", "j3", "gz_seq", "gz_j2")):
    """-> list of (mode, got, want) for the modes that disagree with the reference"""
    path = os.path.join(d, "c.fq" if fastq else "c.fa")
    with open(path, "wb") as f:
        f.write(data)
    want = reference_records(path, unit)
    bad = []
    env_small = dict(os.environ, KU_REGION_KB="1")
    for mode in modes:
        if mode == "seq":
            got = dump_records(["-u", str(unit), path])
        elif mode == "j3":
            got = dump_records(["-u", str(unit), "-j", "3", path])
        else:
            gz = path + ".gz"
            if not os.path.exists(gz):
                with gzip.open(gz, "wb") as f:
                    f.write(data)
            if len(data) == 0:
                continue
            if mode == "gz_seq":
                got = dump_records(["-u", str(unit), gz])
            else:
                r = subprocess.run([DUMP, "-u", str(unit), "-j", "2", gz], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env_small)
                assert r.returncode == 0, r.stderr.decode()
                got = [tuple(ln.split(b"\t", 1)) for ln in r.stdout.split(b"\n")[:-1]]
        if got != want:
            bad.append((mode, got, want))
    for fn in os.listdir(d):
        os.remove(os.path.join(d, fn))
    return bad


def fuzz(n_cases, seed, modes=("seq", "j3", "gz_seq", "gz_j2"), verbose=False):
    rng = random.Random(seed)
    failures = []
    with tempfile.TemporaryDirectory() as d:
        for i in range(n_cases):
            data, fastq, unit = make_case(rng)
            bad = run_case(data, fastq, unit, d, modes)
            if bad:
                failures.append((i, data, fastq, unit, bad))
                if verbose:
                    print(f"case {i} (fastq={fastq}, -u {unit}): {data!r}")
                    for mode, got, want in bad:
                        print(f"   {mode}: got {got}\n   {' ' * len(mode)}  want {want}")
    return failures


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    f = fuzz(n, seed, verbose=True)
    print(f"{n} cases, seed {seed}: {len(f)} disagree with the reference")
    sys.exit(1 if f else 0)
