python3 -m pytest tests -m gpu -q -x 2>&1 | grep -v "^Extension\|^  File" | tail -6
