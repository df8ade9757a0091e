/*
 * ply_oracle.c -- CPU restatement of the reference's Stanford-PLY reader.
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  PARITY UNPINNED.
 *
 * Follows sceneObjects/TriangleMesh.java:23-94 (header scan), :128-153 (readASCII) and
 * :155-197 (readBinary: DataInputStream => big-endian, every vertex property a 4-byte float,
 * 1-byte list count ignored), including its quirks:
 *   - property slots are counted over every non-list "property" line of the header, whichever
 *     element it belongs to (TriangleMesh.java:48-65);
 *   - slots are matched by line suffix " nx"/" ny"/" nz" before " x"/" y"/" z";
 *   - ASCII lines are split on single spaces (String.split(" ")): repeated spaces yield empty
 *     tokens, which then fail to parse (the reference would throw and exit);
 *   - ASCII faces are read until end of file and take tokens 1..3 of each line.
 */
#include "rt_oracle.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_ply_load(const char *path, uint32_t *n_vertices, uint32_t *n_faces, float **vertices, float **normals,
                 int32_t **indices, int32_t *has_normals);
void orc_ply_free(void *p);

/* BufferedReader.readLine: line ends at \n, \r or \r\n; returns 0 at EOF with nothing read */
static int read_line(FILE *f, char **buf, size_t *cap)
{
   size_t n = 0;
   int c = fgetc(f);
   if (c == EOF) return 0;
   for (;;) {
      if (c == EOF || c == '\n') break;
      if (c == '\r') {
         int d = fgetc(f);
         if (d != '\n' && d != EOF) ungetc(d, f);
         break;
      }
      if (n + 2 > *cap) {
         *cap = *cap ? *cap * 2 : 256;
         *buf = (char *) realloc(*buf, *cap);
      }
      (*buf)[n++] = (char) c;
      c = fgetc(f);
   }
   if (n + 1 > *cap) {
      *cap = *cap ? *cap * 2 : 256;
      *buf = (char *) realloc(*buf, *cap);
   }
   (*buf)[n] = 0;
   return 1;
}

static int starts_with(const char *s, const char *p) { return strncmp(s, p, strlen(p)) == 0; }
static int ends_with(const char *s, const char *p)
{
   size_t ls = strlen(s), lp = strlen(p);
   return ls >= lp && strcmp(s + ls - lp, p) == 0;
}

/* String.split(" "): tokens between single spaces; trailing empty tokens dropped */
static int split_spaces(char *line, char **tok, int max_tok)
{
   int n = 0;
   char *p = line;
   for (;;) {
      char *sp = strchr(p, ' ');
      if (n < max_tok) tok[n] = p;
      n++;
      if (!sp) break;
      *sp = 0;
      p = sp + 1;
   }
   if (n > max_tok) n = max_tok;
   while (n > 0 && tok[n - 1][0] == 0) n--;
   return n;
}

static int parse_float(const char *s, float *out)
{
   char *end;
   if (!s || !*s) return 0;
   errno = 0;
   *out = strtof(s, &end);
   return *end == 0;
}
static int parse_int(const char *s, int32_t *out)
{
   char *end;
   long v;
   if (!s || !*s || *s == ' ') return 0;
   errno = 0;
   v = strtol(s, &end, 10);
   if (*end != 0 || errno) return 0;
   *out = (int32_t) v;
   return 1;
}

static int read_be32(FILE *f, uint32_t *out)
{
   unsigned char b[4];
   if (fread(b, 1, 4, f) != 4) return 0;
   *out = ((uint32_t) b[0] << 24) | ((uint32_t) b[1] << 16) | ((uint32_t) b[2] << 8) | (uint32_t) b[3];
   return 1;
}

#define MAX_TOK 64

int orc_ply_load(const char *path, uint32_t *n_vertices, uint32_t *n_faces, float **vertices, float **normals,
                 int32_t **indices, int32_t *has_normals)
{
   FILE *f = fopen(path, "rb");
   char *line = NULL;
   size_t cap = 0;
   int body = 0, is_ascii = 1, property_count = 0;
   int xpos = -1, ypos = -1, zpos = -1, nxpos = -1, nypos = -1, nzpos = -1;
   int32_t nv = 0, nf = 0;
   float *V = NULL, *N = NULL;
   int32_t *I = NULL;
   int rc = -1, hasn;
   char *tok[MAX_TOK];
   if (!f) return -1;

   while (!body && read_line(f, &line, &cap)) {
      if (starts_with(line, "element vertex")) {
         char tmp[256];
         strncpy(tmp, line, sizeof(tmp) - 1);
         tmp[sizeof(tmp) - 1] = 0;
         if (split_spaces(tmp, tok, MAX_TOK) < 3 || !parse_int(tok[2], &nv) || nv < 0) goto done;
      } else if (starts_with(line, "element face")) {
         char tmp[256];
         strncpy(tmp, line, sizeof(tmp) - 1);
         tmp[sizeof(tmp) - 1] = 0;
         if (split_spaces(tmp, tok, MAX_TOK) < 3 || !parse_int(tok[2], &nf) || nf < 0) goto done;
      } else if (starts_with(line, "property")) {
         if (ends_with(line, " nx")) nxpos = property_count;
         else if (ends_with(line, " ny")) nypos = property_count;
         else if (ends_with(line, " nz")) nzpos = property_count;
         else if (ends_with(line, " x")) xpos = property_count;
         else if (ends_with(line, " y")) ypos = property_count;
         else if (ends_with(line, " z")) zpos = property_count;
         if (!starts_with(line, "property list")) property_count++;
      } else if (starts_with(line, "format binary")) {
         is_ascii = 0;
      } else if (strcmp(line, "end_header") == 0) {
         body = 1;
      }
   }
   hasn = (nxpos != -1) && (nypos != -1) && (nzpos != -1);
   if (xpos < 0 || ypos < 0 || zpos < 0) goto done; /* the reference would index with -1 and throw */
   V = (float *) calloc((size_t) (nv ? nv : 1) * 3, sizeof(float));
   N = (float *) calloc((size_t) (nv ? nv : 1) * 3, sizeof(float));
   I = (int32_t *) calloc((size_t) (nf ? nf : 1) * 3, sizeof(int32_t));

   if (is_ascii) {
      int32_t i, ctr = 0;
      for (i = 0; i < nv; i++) {
         int nt;
         if (!read_line(f, &line, &cap)) goto done;
         nt = split_spaces(line, tok, MAX_TOK);
         if (xpos >= nt || ypos >= nt || zpos >= nt) goto done;
         if (!parse_float(tok[xpos], &V[3 * i]) || !parse_float(tok[ypos], &V[3 * i + 1]) || !parse_float(tok[zpos], &V[3 * i + 2])) goto done;
         if (hasn) {
            if (nxpos >= nt || nypos >= nt || nzpos >= nt) goto done;
            if (!parse_float(tok[nxpos], &N[3 * i]) || !parse_float(tok[nypos], &N[3 * i + 1]) || !parse_float(tok[nzpos], &N[3 * i + 2])) goto done;
         }
      }
      while (read_line(f, &line, &cap)) {
         int nt = split_spaces(line, tok, MAX_TOK);
         if (ctr >= nf || nt < 4) goto done; /* ArrayIndexOutOfBounds in the reference */
         if (!parse_int(tok[1], &I[3 * ctr]) || !parse_int(tok[2], &I[3 * ctr + 1]) || !parse_int(tok[3], &I[3 * ctr + 2])) goto done;
         ctr++;
      }
      /* fewer face lines than declared leave zero-filled faces, as in the reference */
   } else {
      /* readBinary re-opens the file and skips the header byte-wise up to the line that starts with end_header */
      int32_t i;
      int j;
      float *props;
      rewind(f);
      for (;;) {
         size_t n = 0;
         int c;
         if (cap < 16) { cap = 256; line = (char *) realloc(line, cap); }
         while ((c = fgetc(f)) != '\n') {
            if (c == EOF) goto done;
            if (n + 2 > cap) { cap *= 2; line = (char *) realloc(line, cap); }
            line[n++] = (char) c;
         }
         line[n] = 0;
         if (starts_with(line, "end_header")) break;
      }
      props = (float *) calloc((size_t) (property_count ? property_count : 1), sizeof(float));
      for (i = 0; i < nv; i++) {
         for (j = 0; j < property_count; j++) {
            uint32_t u;
            if (!read_be32(f, &u)) { free(props); goto done; }
            memcpy(&props[j], &u, 4);
         }
         V[3 * i] = props[xpos];
         V[3 * i + 1] = props[ypos];
         V[3 * i + 2] = props[zpos];
         if (hasn) {
            N[3 * i] = props[nxpos];
            N[3 * i + 1] = props[nypos];
            N[3 * i + 2] = props[nzpos];
         }
      }
      free(props);
      for (i = 0; i < nf; i++) {
         uint32_t a, b, c2;
         if (fgetc(f) == EOF) goto done;
         if (!read_be32(f, &a) || !read_be32(f, &b) || !read_be32(f, &c2)) goto done;
         I[3 * i] = (int32_t) a;
         I[3 * i + 1] = (int32_t) b;
         I[3 * i + 2] = (int32_t) c2;
      }
   }
   *n_vertices = (uint32_t) nv;
   *n_faces = (uint32_t) nf;
   *vertices = V;
   *normals = N;
   *indices = I;
   *has_normals = hasn;
   V = N = NULL;
   I = NULL;
   rc = 0;
done:
   free(V);
   free(N);
   free(I);
   free(line);
   fclose(f);
   return rc;
}

void orc_ply_free(void *p) { free(p); }
