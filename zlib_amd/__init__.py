"""MI355X-native DEFLATE/zlib compressor: Python host mirror over the C ABI."""
from .api import (CORPUS_KINDS, Context, HuffmanTree, Literal, LZ77Compressor, Marker, Strategy, ZgError,
                  adler32, adler32_combine, adler32_device, adler32_device_noalloc, compress, compress_batch, compress_batch_bound, compress_batch_device, compress_bound,
                  compress_device, compress_file, compress_multi, compress_stream, corpus, decompress, decompress_batch, decompress_blocks_device, corpus_into,
                  default_context, last_block_offsets, pack_units, shard_begin, shard_finish, shard_merge, shard_next, shard_open, shard_seal, unpack_units, zlib_compressor)

__all__ = [
    "CORPUS_KINDS", "Context", "HuffmanTree", "Literal", "LZ77Compressor", "Marker", "Strategy", "ZgError", "adler32",
    "adler32_combine", "adler32_device", "adler32_device_noalloc", "compress", "compress_batch", "compress_batch_bound", "compress_batch_device", "compress_bound",
    "compress_device", "compress_file", "compress_multi", "compress_stream", "decompress", "decompress_batch", "decompress_blocks_device", "corpus", "corpus_into", "default_context",
    "last_block_offsets", "pack_units", "shard_begin", "shard_finish", "shard_merge", "shard_next", "shard_open", "shard_seal", "unpack_units", "zlib_compressor",
]
