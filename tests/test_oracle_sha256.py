"""Pin the SHA-256 oracle: FIPS 180-4 known answers + OpenSSL via hashlib (the library the
reference calls, src/hashing_perf/hash.cpp:35)."""
import hashlib

from conftest import seeded_block

FIPS = {
    b"abc": "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad",
    b"": "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855",
    b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq":
        "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1",
}


def test_fips_vectors(oracle):
    for m, want in FIPS.items():
        assert oracle.sha256(m).hex() == want


def test_against_openssl(oracle):
    for n in list(range(0, 130)) + [4095, 4096, 4097, 65536, 100000]:
        d = seeded_block(n, n, "random")
        assert oracle.sha256(d) == hashlib.sha256(d).digest(), n
