"""Corpus for tools/host_asan/harness.cpp: prediction-CSV-shaped files, well-formed and adversarial."""
import os, sys
import numpy as np
out = sys.argv[1] if len(sys.argv) > 1 else 'corpus'
os.makedirs(out, exist_ok=True)
rng = np.random.default_rng(0)
hdr = 'scorer,m,m,m\nbodyparts,a,a,a\ncoords,x,y,likelihood\n'
def w(name, data, binary=False):
    with open(os.path.join(out, name), 'wb') as f:
        f.write(data if binary else data.encode())
def table(n, cols=3, fmt=repr):
    return ''.join(f'{i},' + ','.join(fmt(float(v)) for v in rng.normal(size=cols) * 10.0 ** float(rng.integers(-5, 6))) + '\n' for i in range(n))
w('ok_small.csv', hdr + table(7))
w('ok_large.csv', hdr + table(5000))
w('ok_wide.csv', hdr.replace(',m,m,m', ',m' * 300).replace(',a,a,a', ',a' * 300).replace(',x,y,likelihood', ',x' * 300) + table(50, 300))
w('no_trailing_newline.csv', (hdr + table(9)).rstrip('\n'))
w('crlf.csv', (hdr + table(9)).replace('\n', '\r\n'))
w('empty.csv', '')
w('only_header.csv', hdr)
w('header_cut.csv', hdr[:25])
w('blank_lines.csv', hdr + '\n\n' + table(3) + '\n\n' + table(2) + '\n')
w('ragged_short.csv', hdr + table(5) + '3,1.0\n' + table(2))
w('ragged_long.csv', hdr + table(5) + '3,1.0,2.0,3.0,4.0,5.0\n' + table(2))
w('text_field.csv', hdr + table(5) + '9,abc,1,2\n')
w('na_tokens.csv', hdr + '0,NaN,nan,NA\n1,,N/A,null\n2,-inf,inf,+Infinity\n3,#N/A,<NA>,1\n')
w('long_digits.csv', hdr + '0,' + '1' * 400 + ',0.' + '0' * 350 + '7,' + '9' * 30 + '.5e-400\n' + '1,1e999,-1e-999,1e+\n')
w('signs.csv', hdr + '0,+1.5,-.5,5.\n1,1e5,1E-5,+1e+5\n2,-,+,.\n3,e5,1e,--1\n')
w('nul_bytes.csv', (hdr + table(3)).encode() + b'4,1.0,\x002.0,3.0\n' + table(2).encode(), binary=True)
w('high_bytes.csv', (hdr + table(3)).encode() + b'4,1.0,\xff\xfe,3.0\n', binary=True)
w('truncated_mid_field.csv', (hdr + table(20))[:-7])
w('huge_ints.csv', hdr + '0,9007199254740993,1,2\n18446744073709551616,1,2,3\n')
w('quotes.csv', hdr + '0,"1.5",2,3\n1,"a,b",2,3\n')
w('spaces.csv', hdr + '0, 1.5 ,2 ,3\n1,\t2,3,4\n')
w('one_column.csv', 'a\nb\nc\n1\n2\n3\n')
w('commas_only.csv', hdr + ',,,\n,,,\n')
w('random_bytes.csv', rng.integers(0, 256, 5000, dtype=np.uint8).tobytes(), binary=True)
w('random_csvish.csv', bytes(rng.choice(list(b'0123456789.,-+eE\n \r"naNif'), 20000).astype(np.uint8)), binary=True)
print(len(os.listdir(out)), 'files in', out)
