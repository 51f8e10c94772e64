"""Synthetic workloads of BASELINE.md §4 (numpy only; deterministic counter-based PRNG = numpy Philox).

Every generator returns (chars, lens): chars is a (B, stride) uint8 array with stride % 16 == 0, lens a (B,)
uint32 array.  Bytes stay inside the 98-symbol alphabet {9,10,13,32..126} of the reference's test DFAs
(anything else makes the reference panic, src/lib.rs:817) unless a generator says otherwise.
"""
import numpy as np

ALPHABET98 = np.array([9, 10, 13] + list(range(32, 127)), dtype=np.uint8)
LOWER = np.arange(ord("a"), ord("z") + 1, dtype=np.uint8)


def _rng(seed, stream=0):
    return np.random.Generator(np.random.Philox(key=[seed, stream]))


def _stride(n):
    return max(16, (int(n) + 15) // 16 * 16)


def noise(B, n, seed=0, alphabet=ALPHABET98, stride=None):
    """i.i.d. uniform bytes over `alphabet`, every string exactly n bytes."""
    rng = _rng(seed)
    stride = _stride(n) if stride is None else stride
    chars = np.zeros((B, stride), np.uint8)
    chars[:, :n] = alphabet[rng.integers(0, len(alphabet), size=(B, n), dtype=np.int64)]
    return chars, np.full(B, n, np.uint32)


def _plant(chars, lens, rng, make_text):
    B = chars.shape[0]
    for b in range(B):
        t = make_text(rng)
        n = int(lens[b])
        if len(t) > n:
            continue
        off = int(rng.integers(0, n - len(t) + 1))
        chars[b, off:off + len(t)] = np.frombuffer(t, np.uint8)
    return chars


def _word(rng, lo, hi):
    return bytes(LOWER[rng.integers(0, 26, size=int(rng.integers(lo, hi + 1)))])


def regex1_planted(B, n, seed=0, stride=None):
    """cfg 2(b): noise with `email was meant for @<1-4 lowercase>.` planted at a uniform offset
    (the literal of test_regexes/regex1_test.json) so that tagging, flags and masks are exercised."""
    chars, lens = noise(B, n, seed, stride=stride)
    rng = _rng(seed, 1)
    return _plant(chars, lens, rng, lambda r: b"email was meant for @" + _word(r, 1, 4) + b"."), lens


def regex23_planted(B, n, seed=1, stride=None):
    """cfg 3: noise with ` Also for <word>.` (regex2) and `\\r\\nfrom:<name><user@host>\\r\\n` (regex3) planted."""
    chars, lens = noise(B, n, seed, stride=stride)
    rng = _rng(seed, 1)
    _plant(chars, lens, rng, lambda r: b" Also for " + _word(r, 1, 8) + b".")
    _plant(chars, lens, rng, lambda r: b"\r\nfrom:" + _word(r, 1, 6) + b"<" + _word(r, 1, 6) + b"@" + _word(r, 1, 6)
           + b".com>\r\n")
    return chars, lens


def headers_planted(B, n, seed=3, stride=None):
    """cfg 4: noise with one e-mail-header block planted per string — a `from:`, a `to:` and a `subject:Send <amount>
    <TOKEN> to <address>` line — for the D=3 stand-in definitions tests/golden/dfa/header_{from,to,subject}.json
    (1 / 1 / 3 public parts; compiled by this repo's regex compiler, BASELINE.md §4 cfg 4)."""
    chars, lens = noise(B, n, seed, stride=stride)
    rng = _rng(seed, 1)
    up = lambda r, lo, hi: _word(r, lo, hi).upper()

    def block(r):
        addr = lambda: _word(r, 1, 6) + b"@" + _word(r, 1, 5) + b"." + _word(r, 2, 3)
        amount = str(int(r.integers(1, 100000))).encode() + (b"." + str(int(r.integers(0, 100))).encode() if r.integers(0, 2) else b"")
        return (b"\r\nfrom:" + _word(r, 1, 6) + b" <" + addr() + b">\r\nto:" + addr() + b"\r\nsubject:Send " + amount + b" "
                + up(r, 2, 4) + b" to " + addr() + b"\r\n")
    return _plant(chars, lens, rng, block), lens


def ragged(B, n_max, seed=5, alphabet=ALPHABET98, planted=True):
    """Variable lengths 0..n_max (including empty and full strings), regex1-style plants."""
    rng = _rng(seed, 2)
    stride = _stride(n_max)
    chars = np.zeros((B, stride), np.uint8)
    lens = rng.integers(0, n_max + 1, size=B).astype(np.uint32)
    if B >= 2:
        lens[0], lens[1] = 0, n_max
    body = alphabet[rng.integers(0, len(alphabet), size=(B, stride), dtype=np.int64)]
    for b in range(B):
        chars[b, :lens[b]] = body[b, :lens[b]]
        chars[b, lens[b]:] = 0xAA  # garbage past the end must never be looked at
    if planted:
        _plant(chars, lens, rng, lambda r: b"email was meant for @" + _word(r, 1, 4) + b".")
    return chars, lens


def reveal_stress(B, n, seed=7):
    """Strings built from the pieces that drive the reveal-mask scans of regex1/2/3 through every branch:
    complete matches, matches cut before their end (start_mask set, never reset), long public parts crossing
    64-row tile boundaries, back-to-back matches."""
    rng = _rng(seed, 3)
    stride = _stride(n)
    chars = np.zeros((B, stride), np.uint8)
    lens = np.zeros(B, np.uint32)
    pieces = [
        lambda r: b"email was meant for @" + _word(r, 1, 4) + b".",
        lambda r: b"email was meant for @" + _word(r, 1, 4),             # no closing '.'
        lambda r: b"email was meant for @" + _word(r, 30, 90) + b".",     # public part crosses tiles
        lambda r: b" Also for " + _word(r, 1, 20) + b".",
        lambda r: b" Also for " + _word(r, 60, 150),                      # long and unterminated
        lambda r: b"from:" + _word(r, 1, 9) + b"@" + _word(r, 1, 9) + b".com\r\n",
        lambda r: b"from:" + _word(r, 1, 9) + b"<" + _word(r, 1, 30) + b"@" + _word(r, 1, 9) + b".com>\r\n",
        lambda r: b"from:" + _word(r, 1, 9) + b"<" + _word(r, 40, 100),  # tagged run without an end flag
        lambda r: bytes(ALPHABET98[r.integers(0, 98, size=int(r.integers(1, 80)))]),
    ]
    for b in range(B):
        out = b""
        target = int(rng.integers(0, n + 1))
        while len(out) < target:
            out += pieces[int(rng.integers(0, len(pieces)))](rng)
        out = out[:target]
        chars[b, :len(out)] = np.frombuffer(out, np.uint8)
        lens[b] = len(out)
    return chars, lens


def random_dfa_multi(n_states, seed=2, alphabet=ALPHABET98, n_substr_pairs=40, n_substrs=2):
    """The total DFA of random_dfa(n_states, seed, alphabet) with `n_substrs` substring definitions instead of one (SURVEY §8d cfg 5: "1-2 substr defs drawn as
    random transition subsets"): n_substr_pairs tagged (state, next) pairs EACH, disjoint, so that every definition's id turns up in the rows.
    Returns (allstr_text, [substr_text, ...])."""
    allstr, _ = random_dfa(n_states, seed=seed, alphabet=alphabet, n_substr_pairs=1)
    pairs = sorted({(int(l.split()[0]), int(l.split()[1])) for l in allstr.splitlines()[3:]})
    rng = _rng(seed, 10)
    order = rng.permutation(len(pairs))
    subs = []
    for j in range(n_substrs):
        pick = [pairs[i] for i in order[j * n_substr_pairs:(j + 1) * n_substr_pairs]]
        starts = sorted({a for a, _ in pick[: max(1, len(pick) // 4)]})
        ends = sorted({b for _, b in pick[len(pick) // 2:]})
        subs.append("\n".join(["16", "0", "1023", " ".join(map(str, starts)) + " ", " ".join(map(str, ends)) + " "] + ["%d %d" % p for p in sorted(pick)]) + "\n")
    return allstr, subs


def random_dfa(n_states, seed=2, alphabet=ALPHABET98, n_substr_pairs=40, total=True):
    """A synthetic total (or, with total=False, 90%-dense) DFA over `alphabet` in the reference's text format
    (src/defs.rs:60-68) plus one substring definition (defs.rs:165-177) drawn from its transitions.
    Returns (allstr_text, substr_text)."""
    rng = _rng(seed, 9)
    nxt = rng.integers(0, n_states, size=(n_states, len(alphabet)))
    lines = ["0", str(int(rng.integers(0, n_states))), str(n_states - 1)]
    pairs = set()
    for s in range(n_states):
        for k, ch in enumerate(alphabet):
            if total or rng.random() < 0.9 or s == 0:
                lines.append("%d %d %d" % (s, int(nxt[s, k]), int(ch)))
                pairs.add((s, int(nxt[s, k])))
    pairs = sorted(pairs)
    pick = [pairs[i] for i in rng.choice(len(pairs), size=min(n_substr_pairs, len(pairs)), replace=False)]
    starts = sorted({a for a, _ in pick[: max(1, len(pick) // 4)]})
    ends = sorted({b for _, b in pick[len(pick) // 2:]})
    sub = ["16", "0", "1023", " ".join(map(str, starts)) + " ", " ".join(map(str, ends)) + " "] + ["%d %d" % p for p in sorted(pick)]
    return "\n".join(lines) + "\n", "\n".join(sub) + "\n"
