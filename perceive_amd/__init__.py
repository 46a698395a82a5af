"""perceive_amd — MI355X-native embedding encode + exact similarity scan behind perceive-core's
Model / Searcher API (dimfeld/perceive).  Product code: HIP kernels + C ABI in csrc/, this package
is the thin host mirror.  Nothing here imports oracle/ (test infrastructure)."""
from ._ffi import PcvError, LIB_PATH  # noqa: F401
from .context import Context, device_count  # noqa: F401
from .search import (  # noqa: F401
    SearchItem,
    Searcher,
    cosine_similarity_multi_query,
    cosine_similarity_single_query,
    deserialize_embedding,
    dot_product,
    encode_query,
    merge_topk,
    serialize_embedding,
)
from .sharded import HIT_DTYPE, NativeComm, ShardedSearcher, merge_topk_host, shard_bounds  # noqa: F401,E402
from .model import (  # noqa: F401,E402
    Model,
    ModelError,
    SentenceEmbeddingsModelType,
    make_desc,
    minilm_l6_desc,
    save_weights,
)
from .tokenizer import AlbertTokenizer, BertTokenizer, RobertaTokenizer, TokenizedInput, nfkc  # noqa: F401,E402
from .database import (  # noqa: F401,E402
    Database,
    Item,
    ItemMetadata,
    EMBEDDING_BATCH_SIZE,
    build_searcher,
    calculate_embeddings,
    load_searcher_cache,
    rebuild_source,
    save_searcher_cache,
    search_and_retrieve,
    search_vector_and_retrieve,
)
from .pretrained import checkpoint_tensors, new_pretrained, parse_model_dir  # noqa: F401,E402
